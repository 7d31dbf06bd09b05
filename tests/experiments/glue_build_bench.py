"""CREATE INDEX through the reference's own glue (embedding.c on the mini-Postgres of oracle/pgmock):
the patched glue + libembedding_gpuc.so + hnsw_gpu_server (rows stored by the table scan, one device
build, link lists written back into the pages) next to the reference's objects on the host CPU
(row-by-row hnsw_bind_point).  Usage: python tests/experiments/glue_build_bench.py [rows dims m efconstruction]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import server_util as SU                                   # noqa: E402
from pg_embedding_amd.server import ServerProcess          # noqa: E402
import test_pg_glue as T                                   # noqa: E402

n, dim, m, efc = (int(x) for x in (sys.argv[1:5] + ["100000", "128", "16", "64"][len(sys.argv) - 1:]))
with_ref = "noref" not in sys.argv[5:]                     # the reference's row-by-row build takes ~1 ms per row and more
nq = 8 if n <= 200000 else 0                                # the exact sequential scans cost one DIST request per row
script = T.build_script(n, dim, nq, f"dims={dim},m={m},efconstruction={efc},efsearch=64")
head = "\n".join(script.splitlines()[:4]) + "\n"
exe = SU.build_pg_regress("patched")
for batch in ("0", "1"):
    if batch == "1" and n > 30000:
        continue
    with ServerProcess() as s:
        r = subprocess.run([exe], input=script if batch == "0" else head, capture_output=True, text=True, timeout=3000,
                           env=dict(os.environ, PG_EMBEDDING_GPU_SERVER=s.socket_path, PG_EMBEDDING_GPU_BUILD_BATCH=batch))
    assert r.returncode == 0, r.stderr[-2000:]
    ms = float(re.search(r"Time: ([0-9.]+) ms  create_index", r.stderr).group(1))
    extra = ""
    if batch == "0" and nq:
        res = T.ids_by_statement(r.stdout)
        hits = sum(len(set(res[f"ann {i}"]) & set(res[f"exact {i}"])) for i in range(nq))
        extra = f", recall@10 of the scans that follow {hits / (10 * nq):.3f}"
    print(f"patched glue + server, {'batched device build' if batch == '0' else 'serial device build (bit-identical graph)'}: "
          f"CREATE INDEX {n} x {dim} m={m} efconstruction={efc}: {ms / 1e3:.2f} s{extra}", flush=True)
if with_ref:
    rr = subprocess.run([SU.PG_REGRESS_REF], input=head, capture_output=True, text=True, timeout=3000)
    ref_ms = float(re.search(r"Time: ([0-9.]+) ms  create_index", rr.stderr).group(1))
    print(f"reference glue + hnswalg.o + distfunc.o on one host core: CREATE INDEX {n} x {dim}: {ref_ms / 1e3:.2f} s")
