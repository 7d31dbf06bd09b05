"""Race detection for hnsw_gpu_server: the server's source + the CPU engine double, built with
-fsanitize=thread, then tests/test_server_cpu.py against that binary.  Not collected by pytest (slow,
needs libtsan).  Usage: python tests/experiments/server_tsan.py   -> exit status 0 and no
"WARNING: ThreadSanitizer" on stderr."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = "/tmp/hgs_tsan"
os.makedirs(OUT, exist_ok=True)
INC = ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "pg_embedding_amd", "csrc")]
run = lambda cmd: subprocess.run(cmd, check=True)
run(["gcc", "-O1", "-g", "-fsanitize=thread", "-std=gnu11"] + INC + ["-c", os.path.join(ROOT, "tests/double/engine_double.c"), "-o", OUT + "/double.o"])
run(["gcc", "-O1", "-g", "-fsanitize=thread", "-mavx2", "-mfma", "-ffp-contract=off", "-std=gnu11"] + INC +
    ["-c", os.path.join(ROOT, "oracle/hnsw_port.c"), "-o", OUT + "/port.o"])
run(["g++", "-O1", "-g", "-fsanitize=thread", "-std=c++17"] + INC + [os.path.join(ROOT, "pg_embedding_amd/csrc/server_main.cpp"),
     OUT + "/double.o", OUT + "/port.o", "-o", OUT + "/server_tsan", "-lpthread", "-lm"])
driver = f"""
import sys
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); sys.path.insert(0, {ROOT!r})
import pytest, server_util as SU
SU.build_double_server = lambda: {OUT + '/server_tsan'!r}
sys.exit(pytest.main([{os.path.join(ROOT, 'tests/test_server_cpu.py')!r}, "-x", "-q", "-p", "no:cacheprovider",
                      "-k", "not refuses_to_start and not fails_loudly"]))
"""
r = subprocess.run([sys.executable, "-c", driver], env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0"),
                   capture_output=True, text=True)
races = r.stderr.count("WARNING: ThreadSanitizer")
print(r.stdout[-300:])
print(f"ThreadSanitizer reports: {races}")
sys.exit(1 if (r.returncode or races) else 0)
