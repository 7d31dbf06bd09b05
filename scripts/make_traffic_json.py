"""Turn the FETCH_SIZE / WRITE_SIZE PMC passes of scripts/profile_bench.sh into profiles/traffic.json."""
import csv, glob, hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_digest():
    h = hashlib.sha256()
    for f in ("device_dist.h", "device_search.h"):
        with open(os.path.join(ROOT, "pg_embedding_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def last_mean(d, counter, last=3, pat="hnsw_search"):
    vals = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter and pat in r["Kernel_Name"]:
                    vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    vals.sort()
    tail = [v for _, v in vals[-last:]]
    return sum(tail) / len(tail)

out = sys.argv[1]
fetch_kb = last_mean(os.path.join(out, "pmc_FETCH_SIZE"), "FETCH_SIZE")
write_kb = last_mean(os.path.join(out, "pmc_WRITE_SIZE"), "WRITE_SIZE")
line = json.loads(open(os.path.join(out, "bench_line.json")).read())
cfg = line["config"]
wl = {"n": cfg["rows"], "dim": cfg["dims"], "m": cfg["m"], "ef": cfg["efsearch"], "nq": cfg["queries_per_step_per_gpu"],
      "efc": int(cfg["workload"].split("efconstruction=")[1].split()[0]), "metric": cfg["workload"].split(", ")[2] if False else None}
wl["metric"] = "l2" if ", l2," in cfg["workload"] else ("cosine" if ", cosine," in cfg["workload"] else "manhattan")
print(json.dumps({
    "run": os.path.basename(os.path.normpath(out)),
    "kernel": line["roofline"].get("kernel"),
    "kernel_source_digest": kernel_source_digest(),
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), last 3 dispatches of hnsw_search_kernel of bench.py",
    "FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb,
    "correction": "gfx950: FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads -> doubled (MI355X_MICROARCH.md, HBM section)",
    "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
    "alg_bytes_per_launch": line["roofline"]["alg_bytes_per_launch"],
    "kernel_ms_per_launch_of_that_run": line["roofline"]["kernel_ms_per_launch"],
    "workload": wl}, indent=1))
