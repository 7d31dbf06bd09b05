"""The reference's own scan to exhaustion on a large index: embedding.c UNMODIFIED on the mini-Postgres, a table of
ROWS rows, one ordered scan without LIMIT — hnsw_gettuple doubles efSearch (embedding.c:329-343) until a search comes back short,
so the beams reach the index size — once with the reference's hnswalg.o + distfunc.o underneath and once with libembedding_gpu.so
(in-process device; the wide-beam form, csrc/device_search_wide.h, takes over above efsearch 2048).  The two transcripts must be
the same bytes.

    python tests/experiments/glue_exhaust.py [rows=200000] [dims=16] [--timeout S]
"""
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm(default_seconds=2400.0)      # --timeout SECONDS: a hung device run costs one case, not the round
import server_util as SU                                  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dims = int(sys.argv[2]) if len(sys.argv) > 2 else 16
script = f"""seqscan off
create_table t serial
generate t {rows} {dims} 7
create_index t t_l2 l2 dims={dims},m=8,efconstruction=32,efsearch=64
select t <-> @17 id 0 ; the whole table in index order: efSearch doubles until the index is exhausted
"""


def run(exe, env=None):
    t0 = time.time()
    r = subprocess.run([exe], input=script, capture_output=True, text=True, env=env, timeout=2300)
    return r, time.time() - t0


if not os.path.exists(SU.PG_REGRESS_REF):
    sys.exit("the reference-linked driver (oracle/_ref/pg_regress_ref) is not here: run __graft_entry__.build() where /root/reference exists")
print(f"rows {rows} dims {dims}: reference objects underneath ...", flush=True)
ref, t_ref = run(SU.PG_REGRESS_REF)
print(f"  rc {ref.returncode}, {t_ref:.1f} s, {ref.stdout.count(chr(10))} lines, sha256 {hashlib.sha256(ref.stdout.encode()).hexdigest()[:16]}", flush=True)
print("libembedding_gpu.so underneath ...", flush=True)
gpu, t_gpu = run(SU.build_pg_regress("gpu"), dict(os.environ, PG_EMBEDDING_GPU_STATS="1"))
print(f"  rc {gpu.returncode}, {t_gpu:.1f} s, {gpu.stdout.count(chr(10))} lines, sha256 {hashlib.sha256(gpu.stdout.encode()).hexdigest()[:16]}", flush=True)
print(gpu.stderr[-1500:])
same = ref.returncode == 0 and gpu.returncode == 0 and ref.stdout == gpu.stdout
print("IDENTICAL TRANSCRIPTS" if same else "TRANSCRIPTS DIFFER")
if not same:
    a, b = ref.stdout.splitlines(), gpu.stdout.splitlines()
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print(f"first difference at line {i}: ref {x!r} | gpu {y!r}")
            break
sys.exit(0 if same else 1)
