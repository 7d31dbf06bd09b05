"""Run in its own process by tests/test_pg_glue.py (it binds the flat host's callbacks to the engine-double build of the drop-in
library, process-wide).  A flat host whose memory is changed BEHIND the library's back — what other backends' INSERTs, a VACUUM, a
REINDEX do to the pages of a real host — between calls of hnsw_search / hnsw_bind_point: the validated mirror cache
(csrc/shim_cache.h) must notice along the walk, patch or re-mirror, and keep answering like the reference algorithm over the
host's current bytes.  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                      # noqa: E402

import oracle                           # noqa: E402
import server_util as SU                # noqa: E402
from pg_embedding_amd.datasets import gmm   # noqa: E402

lib_path = SU.build_shim_double()
dim, m, efc, efs, func = 24, 6, 32, 24, 0
h = oracle.FlatHostIndex(lib_path, dim, m, efc, efs, func)
L = C.CDLL(lib_path)
L.hnsw_gpu_shim_cache_stats.argtypes = [C.POINTER(C.c_uint64)]


def stats():
    v = (C.c_uint64 * 8)()
    L.hnsw_gpu_shim_cache_stats(v)
    return dict(zip(("snapshots", "searches", "search_rounds", "inserts", "insert_rounds", "patched", "fallbacks", "elements_read"), map(int, v)))


X = gmm(4000, dim, k=40, seed=11)
Q = gmm(60, dim, k=40, seed=11, stream=1)
port = oracle.PortIndex(dim, m, efc, efs, func)          # the reference algorithm (canonical arithmetic) = what the engine double runs
checks = 0


def same_answers(tag):
    global checks
    for q in Q:
        got = h.search(q, efs)
        want = port.search(q, efs)[0]
        assert got.size == want.size and (got == want).all(), tag
        checks += 1


port.add(X[:1500], np.arange(1500, dtype=np.uint64) + 100)
h.load_raw(port.raw(), 1500)
same_answers("first contact")
s0 = stats()
assert s0["snapshots"] == 1
# --- other backends insert 400 rows and vacuum 60 (link lists rewritten all over the graph, new elements, flag bits)
port.add(X[1500:1900], np.arange(1500, 1900, dtype=np.uint64) + 100)
for i in range(0, 1900, 31):
    port.set_deleted(i)
h.load_raw(port.raw(), 1900)
same_answers("after foreign inserts and a vacuum")
s1 = stats()
assert s1["snapshots"] == 1 and s1["fallbacks"] == 0 and s1["patched"] > 0, s1      # repaired along the walks, no full re-walk
# --- our own inserts through hnsw_bind_point on top of that (each validates what it will read first), mixed with foreign ones
for r in range(1900, 2100):
    if r % 7 == 0:                                  # a foreign insert lands in between
        port.add(X[r:r + 1], np.array([r + 100], np.uint64))
        h.load_raw(port.raw(), r + 1)
    else:
        h.add(X[r:r + 1], np.array([r + 100], np.uint64))       # append + hnsw_bind_point through the library
        port.add(X[r:r + 1], np.array([r + 100], np.uint64))
    assert (h.raw() == port.raw()).all(), r        # the host's pages hold the reference's graph, byte for byte
same_answers("after mixed inserts")
s2 = stats()
assert s2["snapshots"] == 1 and s2["fallbacks"] == 0, s2
# --- REINDEX: another graph over other rows behind the same parameters (the identity guess fails -> one new full walk)
port2 = oracle.PortIndex(dim, m, efc, efs, func)
port2.add(X[2100:3300], np.arange(1200, dtype=np.uint64) + 7)
h.load_raw(port2.raw(), 1200)
port = port2
same_answers("after a rebuild with other rows")
s3 = stats()
assert s3["snapshots"] == 2, s3
# --- the same first row, fewer rows: the mirror names elements the host no longer has
port3 = oracle.PortIndex(dim, m, efc, efs, func)
port3.add(X[2100:2400], np.arange(300, dtype=np.uint64) + 7)
h.load_raw(port3.raw(), 300)
port = port3
same_answers("after a rebuild that shrank the index")
s4 = stats()
print(json.dumps({"checks": checks, "after_first_contact": s0, "after_foreign_changes": s1, "after_mixed_inserts": s2,
                  "after_rebuild": s3, "after_shrink": s4}))
