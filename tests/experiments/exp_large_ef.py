"""Generic form with its sets in LDS vs in HBM at large ef: parity against the CPU oracle on a small index,
then timing on a 200k x 128 index (HNSW_GPU_LDS_SET_MIN_WAVES picks the form: 1 = LDS as long as one wave
fits, 1000 = always HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import oracle, pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm, gmm_torch
dev = torch.device("cuda", 0)

# parity: 6000 x 32, ef up to "everything"
X = gmm(6000, 32, k=30, seed=5); Q = gmm(12, 32, k=30, seed=5, stream=1)
port = oracle.PortIndex(32, 8, 40, 64, pg.DIST_L2); port.add(X)
meta = pg.make_meta(32, 8, 40, 64, pg.DIST_L2)
ix = pg.GpuIndex.from_flat(meta, port.raw(), 6000)
for ef in (700, 3000, 7000, 20000):
    for mw in ("1", "1000"):
        if mw == "1" and ef > 6000: continue
        os.environ["HNSW_GPU_LDS_SET_MIN_WAVES"] = mw
        L, D, Cn = ix.search(Q, ef)
        W = port.search_many(Q, ef)
        ok = (Cn == W["counts"]).all() and all((L[q, :Cn[q]] == W["labels"][q, :Cn[q]]).all() and
              (D[q, :Cn[q]].view(np.uint32) == W["dists"][q, :Cn[q]].view(np.uint32)).all() for q in range(len(Q)))
        print(f"parity ef={ef:6d} sets in {'LDS' if mw == '1' else 'HBM'}: {'ok' if ok else 'MISMATCH'} (results/query {Cn.mean():.0f})", flush=True)
ix.close()

n, dim = 200000, 128
Xd = gmm_torch(n, dim, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, 16, 100, 128, pg.DIST_L2), n); ix.append_torch(Xd); ix.link(0, n); torch.cuda.synchronize()
Qd = gmm_torch(4096, dim, stream=1, device=dev)
for ef in (600, 1000, 1500, 2500, 4000, 6000, 12000):
    for mw in ("1", "1000"):
        if mw == "1" and ef > 6000: continue
        os.environ["HNSW_GPU_LDS_SET_MIN_WAVES"] = mw
        nq = 4096 if ef <= 2500 else 1024
        out = ix.search_torch(Qd[:nq].contiguous(), ef); torch.cuda.synchronize()
        ms = min((ix.search_torch(Qd[:nq].contiguous(), ef, out=out), ix.last_search_ms())[1] for _ in range(2))
        print(f"ef={ef:6d} sets in {'LDS' if mw == '1' else 'HBM'}: {nq} queries {ms:9.1f} ms  {nq/ms*1e3:9.0f} q/s  slots {ix.last_search_slots()}", flush=True)
