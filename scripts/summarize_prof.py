"""Condense rocprofv3 CSV output into a small text summary for profiles/.

usage: summarize_prof.py <dir> [--last K] [kernel-substring ...]
  kernel stats table (from --stats), then for every matching kernel the LAST K dispatches
  (bench.py's timed launches are the final ones of the process; earlier launches of the
  same kernel belong to the index build) of the kernel trace and of every PMC counter.
"""
import collections
import csv
import glob
import os
import sys


def rows(d, pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        with open(f) as fh:
            out += list(csv.DictReader(fh))
    return out


def main():
    args = sys.argv[1:]
    d = args.pop(0)
    last = 5
    if "--last" in args:
        i = args.index("--last")
        last = int(args[i + 1])
        del args[i:i + 2]
    pats = args or ["hnsw_search"]
    ks = rows(d, "*kernel_stats.csv")
    if ks:
        print("| kernel | calls | total ns | avg ns | % |")
        print("|---|---|---|---|---|")
        for r in sorted(ks, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
            print(f"| {r['Name'][:72]} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {float(r['Percentage']):.2f} |")
    kt = rows(d, "*kernel_trace.csv")
    if kt:
        by = collections.defaultdict(list)
        for r in kt:
            by[r["Kernel_Name"]].append((int(r["Dispatch_Id"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r))
        for k, v in by.items():
            if not any(p in k for p in pats):
                continue
            v.sort()
            tail = v[-last:]
            durs = [x[1] for x in tail]
            r = tail[-1][2]
            print(f"\nkernel trace, last {len(tail)} dispatches of {k[:80]}:")
            print(f"  duration ns: {durs}  mean {sum(durs)/len(durs):.0f}")
            print("  grid %s workgroup %s VGPR %s accumVGPR %s SGPR %s LDS %s scratch %s" % (
                r.get("Grid_Size"), r.get("Workgroup_Size"), r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size")))
    cc = rows(d, "*counter_collection.csv")
    if cc:
        by = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in cc:
            by[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        for k, v in by.items():
            if not any(p in k for p in pats):
                continue
            print(f"\nPMC, last {last} dispatches of {k[:80]}:")
            for c, vals in sorted(v.items()):
                vals.sort()
                tail = [x[1] for x in vals[-last:]]
                print(f"  {c}: mean {sum(tail)/len(tail):.6g}  values {['%.6g' % x for x in tail]}")


if __name__ == "__main__":
    main()
