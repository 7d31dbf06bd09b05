"""Host-side plumbing that keeps a round from being lost: the script watchdog (pg_embedding_amd/watchdog.py) and the rule that a
built artefact is current by the BYTES of its sources, not by their modification times (pg_embedding_amd/build.py)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_script_watchdog_ends_a_process_that_hangs():
    """arm() takes `--timeout S` off the command line; when the limit is reached the process dumps its stacks, asks the library to
    end the launches in flight (nothing is loaded here: it says so or stays silent) and exits with status 124 — whatever the main
    thread is blocked in."""
    code = ("import sys, time\nsys.path.insert(0, %r)\nfrom pg_embedding_amd import watchdog\n"
            "lim = watchdog.arm()\nassert lim == 1.0 and '--timeout' not in sys.argv and sys.argv[1:] == ['a', 'b'], sys.argv\n"
            "time.sleep(60)\n") % ROOT
    t0 = time.time()
    r = subprocess.run([sys.executable, "-c", code, "a", "--timeout", "1", "b"], capture_output=True, text=True, timeout=40)
    assert r.returncode == 124, (r.returncode, r.stderr[-500:])
    assert time.time() - t0 < 20 and "limit reached" in r.stderr and "time.sleep" not in r.stdout


def test_artefacts_are_current_by_content_not_by_mtime(tmp_path):
    sys.path.insert(0, ROOT)
    from pg_embedding_amd import build as b
    src = tmp_path / "a.h"
    src.write_text("int x;\n")
    target = tmp_path / "lib.so"
    target.write_text("binary")
    d = b._digest([str(src)], ["cc", "-O2"])
    assert not b._current(str(target), d)                  # no stamp yet
    b._built(str(target), d)
    assert b._current(str(target), d)
    os.utime(src, (time.time() + 1000, time.time() + 1000))  # a copied tree, a touched file: still the same bytes
    assert b._current(str(target), b._digest([str(src)], ["cc", "-O2"]))
    src.write_text("int y;\n")                              # edited — even with an OLDER mtime than the artefact
    os.utime(src, (1, 1))
    assert not b._current(str(target), b._digest([str(src)], ["cc", "-O2"]))
    assert not b._current(str(target), b._digest([str(src)], ["cc", "-O3"]))      # another command line
    os.remove(target)
    assert not b._current(str(target), d)                  # stamp without artefact


def test_the_chunked_generator_yields_the_bytes_of_the_whole_table_generator():
    """pg_embedding_amd.datasets.gmm_chunks (bench.py: the rows of the reference-built headline graph are uploaded chunk by chunk, so that the
    host never holds the 3 GB table): the same bytes as gmm() for every row, whatever the chunk size — the graph in oracle/_ref/serial_graph_*
    was built over exactly gmm()'s rows."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from pg_embedding_amd.datasets import gmm, gmm_chunks
    X = gmm(3001, 37, k=50, sigma=0.3, seed=42, stream=0)
    for chunk in (1, 777, 3001, 1 << 16):
        parts = list(gmm_chunks(3001, 37, k=50, sigma=0.3, seed=42, stream=0, chunk=chunk))
        assert [a for a, _ in parts] == list(range(0, 3001, chunk))
        Y = np.concatenate([x for _, x in parts])
        assert Y.shape == X.shape and (Y.view(np.uint32) == X.view(np.uint32)).all()
