#!/usr/bin/env python3
"""Per-kernel register / spill / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage.

usage: scripts/resource_usage.py [filter-substring] [unit]
Compiles one translation unit of the library for gfx950 (object only, into /tmp) and prints one line per kernel.  unit: "main"
(hnsw_gpu.hip, default), "sort" (sort_pairs.hip) or 1..5 (search_inst.hip with -DSEARCH_INST_SHAPE=unit: the beam kernels of one
row shape — 1 Shape2x4, 2 Shape4x2, 3 Shape8x2, 4 Shape12x2, 5 Shape2x2; pg_embedding_amd/build.py).
"""
import re, subprocess, sys, tempfile, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = sys.argv[1] if len(sys.argv) > 1 else ""
unit = sys.argv[2] if len(sys.argv) > 2 else "main"
src = {"main": "hnsw_gpu.hip", "sort": "sort_pairs.hip"}.get(unit, "search_inst.hip")
extra = ["-DSEARCH_INST_SHAPE=" + unit] if src == "search_inst.hip" else []
with tempfile.TemporaryDirectory() as td:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                        "-I" + os.path.join(ROOT, "include"), *extra, "-c", os.path.join(ROOT, "pg_embedding_amd/csrc", src),
                        "-o", os.path.join(td, "x.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
cur = None
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(pgemb::SearchArgs\)|pgemb::|void ", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>4s}")
for c in rows:
    if flt in c["name"]:
        print(f"{c['name'][:70]:70s} {c.get('VGPRs',0):5d} {c.get('AGPRs',0):5d} {c.get('SGPRs',0):5d} {c.get('VGPRs Spill',0):6d} "
              f"{c.get('SGPRs Spill',0):6d} {c.get('ScratchSize',0):7d} {c.get('Occupancy',0):4d}")
