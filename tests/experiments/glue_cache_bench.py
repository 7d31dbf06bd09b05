"""The reference's UNMODIFIED glue (embedding.c on oracle/pgmock) over libembedding_gpu.so in process: row-by-row CREATE
INDEX (hnsw_bind_point per row) and index scans (hnsw_search per scan), with the validated mirror cache (shim_cache.h),
with the cache off (a full walk + upload per call, the round-1 behaviour; small index only) and with the reference's own
objects on one host core.  Usage: python tests/experiments/glue_cache_bench.py [rows dims m efconstruction scans]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import server_util as SU                                   # noqa: E402

n, dim, m, efc, nscan = (int(x) for x in (sys.argv[1:6] + ["20000", "128", "16", "64", "200"][len(sys.argv) - 1:]))


def script(rows, scans):
    L = ["create_table t serial", f"generate t {rows} {dim} 12345", f"create_index t t_l2 l2 dims={dim},m={m},efconstruction={efc},efsearch=64",
         "seqscan off"]
    L += [f"select t <-> @{(i * 7919 + 13) % rows} id 10 ; ann {i}" for i in range(scans)]
    return "\n".join(L) + "\n"


def run(exe, rows, scans, env=None):
    r = subprocess.run([exe], input=script(rows, scans), capture_output=True, text=True, timeout=6000,
                       env=dict(os.environ, PGEMB_PRINT_CACHE_STATS="1", PGEMB_TIME_SELECTS="1", **(env or {})))
    assert r.returncode == 0, r.stderr[-2000:]
    build = float(re.search(r"Time: ([0-9.]+) ms  create_index", r.stderr).group(1)) / 1e3
    sel = sorted(float(x) for x in re.findall(r"Time: ([0-9.]+) ms  select", r.stderr))
    stats = [ln for ln in r.stderr.splitlines() if ln.startswith("shim cache") or ln.startswith("shim inserts")]
    return r.stdout, build, (sel[len(sel) // 2] if sel else 0.0), " | ".join(stats)


gpu = SU.build_pg_regress(os.environ.get("PGEMB_GLUE_VARIANT", "gpu"))      # "shimdouble": the CPU engine double (dry run)
out_c, b_c, s_c, st = run(gpu, n, nscan)
print(f"unmodified glue + libembedding_gpu.so, validated cache: CREATE INDEX {n} x {dim} m={m} efc={efc} row by row: {b_c:.2f} s "
      f"({b_c / n * 1e3:.3f} ms per insert); index scan (LIMIT 10, efsearch 64) median {s_c:.3f} ms; {st}", flush=True)
out_r, b_r, s_r, _ = run(SU.PG_REGRESS_REF, n, nscan)
print(f"reference glue + hnswalg.o + distfunc.o on one host core: CREATE INDEX {b_r:.2f} s ({b_r / n * 1e3:.3f} ms per insert); "
      f"index scan median {s_r:.3f} ms; same bytes: {out_c == out_r}", flush=True)
if n <= 5000:      # cache off = a full walk + upload per call: only affordable on a small index
    out_o, b_o, s_o, _ = run(gpu, n, nscan, env={"PG_EMBEDDING_GPU_CACHE": "0"})
    print(f"unmodified glue + libembedding_gpu.so, cache off (full walk per call): CREATE INDEX {b_o:.2f} s "
          f"({b_o / n * 1e3:.3f} ms per insert); index scan median {s_o:.3f} ms; same bytes: {out_o == out_r}", flush=True)
