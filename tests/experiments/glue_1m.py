"""The drop-in symbols through the reference's UNMODIFIED glue at FULL size: hnsw_search (index scans) and hnsw_bind_point (row
inserts) on a 1 000 000 x 768 index, the reference's own objects on one host core against libembedding_gpu.so in process — over the
VERY SAME index pages: the index is built once (patched glue + hnsw_gpu_server: the batched device build, seconds instead of the
half hour of a row-by-row CPU build), its page image is saved (oracle/pgmock `save_index`) and attached by both binaries
(`attach_index`) over the same generated table.  VERDICT r3 weak #4: the 20 000-row comparison is the host's best case (everything
in its caches); this is the size the metric is quoted on.
Usage: python tests/experiments/glue_1m.py [rows dims m efconstruction efsearch scans inserts]"""
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm(default_seconds=1500.0)
import numpy as np                                           # noqa: E402
import server_util as SU                                     # noqa: E402
from pg_embedding_amd.server import ServerProcess            # noqa: E402

n, dim, m, efc, efs, nscan, nins = (int(x) for x in (sys.argv[1:8] + ["1000000", "768", "16", "200", "128", "300", "60"][len(sys.argv) - 1:]))
opts = f"dims={dim},m={m},efconstruction={efc},efsearch={efs}"
tmp = tempfile.mkdtemp(prefix="glue1m_")
idxf = os.path.join(tmp, "t_l2.idx")
head = ["create_table t serial", f"generate t {n} {dim} 12345"]

# 1. one build: the patched glue hands CREATE INDEX to the server (batched device build), the pages are saved
t0 = time.time()
DRY = os.environ.get("GLUE1M_DRY") == "1"                  # CPU dry run of this script: the server and engine doubles of the CPU test tier
with ServerProcess(binary=SU.build_double_server() if DRY else None) as srv:
    r = subprocess.run([SU.build_pg_regress("patched")], input="\n".join(head + [f"create_index t t_l2 l2 {opts}", f"save_index t t_l2 {idxf}"]) + "\n",
                       capture_output=True, text=True, timeout=3000, env=dict(os.environ, PG_EMBEDDING_GPU_SERVER=srv.socket_path))
assert r.returncode == 0 and "SAVE" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
build = float(re.search(r"Time: ([0-9.]+) ms  create_index", r.stderr).group(1)) / 1e3
print(f"# {n} x {dim} m={m} efconstruction={efc}: CREATE INDEX through the patched glue + server {build:.1f} s; page image {os.path.getsize(idxf) / 1e9:.2f} GB "
      f"({time.time() - t0:.0f} s with the table)", flush=True)

# 2. the same session over the same pages, twice
rng = np.random.default_rng(5)
rows = rng.integers(0, n, nscan)
new = (rng.integers(0, 64, (nins, dim)) / 8.0).astype(np.float32)
body = [f"attach_index t t_l2 l2 {opts} {idxf}", "seqscan off"]
body += [f"select t <-> @{int(x)} id 10 ; scan {i}" for i, x in enumerate(rows)]
body += ["insert t {" + ",".join(f"{v:g}" for v in new[i]) + "}" for i in range(nins)]
body += [f"select t <-> @{int(x)} id 10 ; rescan {i}" for i, x in enumerate(rows[:50])]         # after the inserts
script = "\n".join(head + body) + "\n"


def session(exe, label, env=None):
    t1 = time.time()
    r = subprocess.run([exe], input=script, capture_output=True, text=True, timeout=3000,
                       env=dict(os.environ, PGEMB_TIME_SELECTS="1", PGEMB_TIME_INSERTS="1", PGEMB_PRINT_CACHE_STATS="1", **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    sel = [float(x) for x in re.findall(r"Time: ([0-9.]+) ms  select", r.stderr)]
    ins = [float(x) for x in re.findall(r"Time: ([0-9.]+) ms  insert", r.stderr)]
    first, sel = sel[0], sel[1:nscan]
    med = lambda v: sorted(v)[len(v) // 2]
    print(f"{label}: index scan (LIMIT 10, efsearch {efs}) median {med(sel):.3f} ms, mean {sum(sel) / len(sel):.3f}, p90 {sorted(sel)[int(0.9 * len(sel))]:.3f} "
          f"(first scan {first:.1f} ms); insert (hnsw_bind_point, efconstruction {efc}) median {med(ins):.3f} ms, mean {sum(ins) / len(ins):.3f}; "
          f"session {time.time() - t1:.0f} s", flush=True)
    for ln in r.stderr.splitlines():
        if ln.startswith("shim cache") or ln.startswith("shim inserts"):
            print("   " + ln, flush=True)
    return r.stdout


def compare(name, out):
    same = out_ref == out
    print(f"   {name}: same result tables as the reference (every scan, before and after the inserts): {same}", flush=True)
    if not same:
        a, b = out_ref.splitlines(), out.splitlines()
        diff = [i for i in range(min(len(a), len(b))) if a[i] != b[i]]
        print(f"   {len(diff)} of {len(a)} lines differ, first at line {diff[0] if diff else -1}: {a[diff[0]] if diff else ''!r} vs {b[diff[0]] if diff else ''!r}")


out_ref = session(SU.PG_REGRESS_REF, "reference glue + hnswalg.o + distfunc.o, one host core")
out_gpu = session(SU.build_pg_regress("shimdouble" if DRY else "gpu"),
                  "UNMODIFIED glue + libembedding_gpu.so in process (validated mirror cache: every walk is checked against the host's pages)")
compare("in process", out_gpu)
# the trusted mirror: the 49-line maintainer patch gives the mirror an identity + generation, the server keeps ONE mirror for all
# backends and no walk is re-read on the host; the first scan of this fresh process uploads the attached pages
with ServerProcess(binary=SU.build_double_server() if DRY else None) as srv:
    out_srv = session(SU.build_pg_regress("patched"), "PATCHED glue (49 lines) + libembedding_gpuc.so + hnsw_gpu_server (trusted mirror)",
                      env={"PG_EMBEDDING_GPU_SERVER": srv.socket_path})
compare("server", out_srv)
os.remove(idxf)
