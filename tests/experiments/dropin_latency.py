"""End-to-end latency of the drop-in symbol hnsw_search() (attached mirror) for ONE query: wall clock per call vs
the kernel time inside it, and the reference's own CPU code on the same graph."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import oracle, pg_embedding_amd as pg
from pg_embedding_amd._lib import gpu_lib
from pg_embedding_amd.datasets import gmm_torch
dev = torch.device("cuda", 0)
n, dim, m, efc, ef = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 768, 16, 200, 128
X = gmm_torch(n, dim, device=dev)
meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
ix = pg.GpuIndex.empty(meta, n); ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize()
Q = gmm_torch(400, dim, stream=1, device=dev).cpu().numpy()
L = gpu_lib()
lab = np.empty(ef, np.uint64); cnt = np.zeros(1, np.uint32)
def one(q):
    rc = L.hnsw_gpu_search_batch(ix._h, q.ctypes.data, 1, ef, lab.ctypes.data, None, cnt.ctypes.data)
    assert rc == 0
for q in Q[:20]: one(q)
t = time.perf_counter()
kms = 0.0
for q in Q:
    one(q); kms += ix.last_search_ms()
wall = (time.perf_counter() - t) / len(Q)
print(f"hnsw_gpu_search_batch(nq=1), host pointers: {wall*1e6:.0f} us per call wall clock, kernel {kms/len(Q)*1e3:.0f} us", flush=True)
if oracle.have_ref() and n <= 200000:
    pass
