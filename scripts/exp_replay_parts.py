"""Roof of a launch with FEWER queries than resident waves: the launch's own evaluation trace gathered again with 1, 2, 4 waves
per query (hnsw_gpu_replay_roof_parts) at several wave counts — what the memory system gives this trace when `parts` waves share
one walk's rows — next to the search kernel's own time.  usage: exp_replay_parts.py <dim> <m> <metric> [nq,nq,...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nqs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1024").split(",")]
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, int(os.environ.get("EXP_EF", "128"))
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, stream=0, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Qall = gmm_torch(max(nqs), dim, stream=1, device=dev)
row_bytes = (dim + 3) // 4 * 16
lpr = (row_bytes // 16 + 15) // 16
shapes = [(2, 2), (2, 4)] if lpr <= 2 else [(4, 2), (4, 4)] if lpr <= 4 else [(8, 2)] if lpr <= 8 else [(12, 2), (12, 1)]
for nq in nqs:
    Q = Qall[:nq].contiguous()
    out = ix.search_torch(Q, ef, stats=True)
    ms = []
    for _ in range(10):
        ix.search_torch(Q, ef, out=out)
        ms.append(ix.last_search_ms())
    st = out["stats"].cpu().numpy().astype(np.int64)
    cnt = out["counts"].cpu().numpy().astype(np.int64)
    byt = (st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum()
    slots = ix.last_search_slots()
    print(f"dim {dim} m {m} {metric} nq={nq}: search {min(ms):.3f} ms (median {sorted(ms)[5]:.3f}) on {slots} slots = {byt / min(ms) / 1e6:.0f} GB/s alg "
          f"= {byt / min(ms) / 1e6 / 8000:.3f} of 8 TB/s [{ix.last_search_kernel()}]", flush=True)
    cap = int(st[:, 0].max()) + 64
    tr = ix.search_traced_torch(Q, ef, evals_cap=cap)
    torch.cuda.synchronize()
    for kb, rpg in shapes:
        for parts in (1, 2, 4, 8):
            for rs in sorted({slots, 2048, 4096}):
                rms, rby = ix.replay_roof(tr, rs, kb, rpg, parts=parts)
                print(f"   replay <{kb},{rpg}> parts {parts} waves {rs:5d}: {rms:.3f} ms = {rby / rms / 1e6:.0f} GB/s rows = {rby / rms / 1e6 / 8000:.3f} of 8 TB/s; "
                      f"search / replay = {min(ms) / rms:.2f}", flush=True)
