"""The drop-in hnsw_search() with NO attached mirror at full size: a flat host (oracle/flat_host.c) holding a device-built
1M x 768 graph, searched through libembedding_gpu.so's validated cache — wall clock per call, next to the attached-mirror
call (no validation).  (The reference's own code is timed by bench.py, in a process where its symbols cannot be interposed.)
usage: dropin_cache_latency.py [rows] [dims]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd import build as B
from pg_embedding_amd._lib import shim_lib
from pg_embedding_amd.datasets import gmm_torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
m, efc, ef = 16, 200, 128
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, pg.DIST_L2), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
raw = ix.export_flat()
ix.close()
del X
Q = gmm_torch(300, dim, stream=1, device=dev).cpu().numpy()
h = oracle.FlatHostIndex(B.SHIM_LIB, dim, m, efc, ef, pg.DIST_L2)
h.load_raw(raw, n)
L = shim_lib()
L.hnsw_gpu_shim_cache_stats.argtypes = [C.POINTER(C.c_uint64)]


def stats():
    v = (C.c_uint64 * 8)()
    L.hnsw_gpu_shim_cache_stats(v)
    return [int(x) for x in v]


t0 = time.perf_counter()
first = h.search(Q[0], ef)
t_first = time.perf_counter() - t0
for q in Q[1:20]:
    h.search(q, ef)
s0 = stats()
walls = []
outs = []
for q in Q[20:]:
    t0 = time.perf_counter()
    outs.append(h.search(q, ef))
    walls.append((time.perf_counter() - t0) * 1e3)
s1 = stats()
nq = len(walls)
print(f"{n} x {dim}, ef {ef}: hnsw_search with NO attached mirror (validated cache): first call (full walk + upload) {t_first:.2f} s; "
      f"then median {np.median(walls):.3f} ms, mean {np.mean(walls):.3f} ms per call; rounds per search {(s1[2] - s0[2]) / nq:.2f}, "
      f"elements read per search {(s1[7] - s0[7]) / nq:.0f}, full walks {s1[0]}", flush=True)
# attached mirror: no validation
mirror = C.c_void_p()
assert L.hnsw_gpu_shim_snapshot(h.meta, C.byref(mirror)) == 0
assert L.hnsw_gpu_shim_attach(h.meta, mirror) == 0
for q in Q[:20]:
    h.search(q, ef)
wa = []
for i, q in enumerate(Q[20:]):
    t0 = time.perf_counter()
    r = h.search(q, ef)
    wa.append((time.perf_counter() - t0) * 1e3)
    assert (r == outs[i]).all()
print(f"  attached mirror (hnsw_gpu_shim_attach): median {np.median(wa):.3f} ms per call, same answers", flush=True)
L.hnsw_gpu_shim_detach(h.meta)
print("  (the reference's own code on one host core over the same kind of graph: bench.py cpu_baseline.single_thread_qps, 2.1 k q/s = 0.47 ms)")
