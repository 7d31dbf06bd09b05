"""Long differential fuzz through the reference's own glue (see tests/test_pg_glue.py::random_session): for
every seed the session runs with the reference's objects, with libembedding_gpuc.so (un-patched glue) and
with the patched glue, all on the mini-Postgres; the three transcripts must be identical.
Usage: python tests/experiments/glue_fuzz.py [first_seed [count [--device | --cache]]]
       (default: the CPU engine double behind the server; --device: the real server, serial build order;
        --cache: the in-process library's validated mirror cache instead — its own source over the CPU engine double, or with
        --device the product library on the GPU)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import server_util as SU                                   # noqa: E402
from pg_embedding_amd.server import ServerProcess          # noqa: E402
import test_pg_glue as T                                   # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
device = "--device" in sys.argv
bad = runs = 0
if "--cache" in sys.argv:
    exe = SU.build_pg_regress("gpu" if device else "shimdouble")
    tot = [0] * 8
    for seed in range(first, first + count):
        script = T.random_session(seed)
        want = subprocess.run([SU.PG_REGRESS_REF], input=script, capture_output=True, text=True, timeout=900)
        assert want.returncode == 0, (seed, want.stderr[-500:])
        got = subprocess.run([exe], input=script, capture_output=True, text=True, timeout=900, env=dict(os.environ, PGEMB_PRINT_CACHE_STATS="1"))
        runs += 1
        if got.returncode != 0 or got.stdout != want.stdout:
            bad += 1
            print("MISMATCH seed", seed, got.returncode, got.stderr[-300:], flush=True)
        for ln in got.stderr.splitlines():
            if ln.startswith("shim cache:"):
                for i, v in enumerate(int(x) for x in ln.split()[3::2]):
                    tot[i] += v
    print(f"{runs} sessions through the validated cache, {bad} mismatches; snapshots {tot[0]} searches {tot[1]} search rounds {tot[2]} "
          f"inserts {tot[3]} insert rounds {tot[4]} patched {tot[5]} fallbacks {tot[6]} elements read {tot[7]}", flush=True)
    sys.exit(1 if bad else 0)
with ServerProcess(binary=None if device else SU.build_double_server()) as s:
    env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=s.socket_path, PG_EMBEDDING_GPU_BUILD_BATCH="1")
    for seed in range(first, first + count):
        script = T.random_session(seed)
        want = subprocess.run([SU.PG_REGRESS_REF], input=script, capture_output=True, text=True, timeout=900)
        assert want.returncode == 0, (seed, want.stderr[-500:])
        for v in ("client", "patched"):
            got = subprocess.run([SU.build_pg_regress(v)], input=script, capture_output=True, text=True, timeout=900, env=env)
            runs += 1
            if got.returncode != 0 or got.stdout != want.stdout:
                bad += 1
                print("MISMATCH seed", seed, v, got.returncode, got.stderr[-300:], flush=True)
print(f"{runs} sessions, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
