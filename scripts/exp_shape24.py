"""<24,1> whole-row batches vs <12,2> for rows wider than 768 floats, by launch size (HNSW_GPU_SHAPE_24X1 = 1 / 0).
usage: exp_shape24.py <dim> <m> [metric] [--timeout S]     env: EXP_NQS"""
import os
import sys
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m = int(sys.argv[1]), int(sys.argv[2])
metric = sys.argv[3] if len(sys.argv) > 3 else "cosine"
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, 128
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, stream=0, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Qall = gmm_torch(40000, dim, stream=1, device=dev)
for nq in [int(x) for x in os.environ.get("EXP_NQS", "1,16,256,1024,2560,5000,10000,40000").split(",")]:
    ref = None
    for s24 in ("0", "1"):
        os.environ["HNSW_GPU_SHAPE_24X1"] = s24
        ms = []
        if nq == 1:
            for i in range(60):
                out = ix.search_torch(Qall[i:i + 1].contiguous(), ef, stats=True)
                ms.append(ix.last_search_ms())
            out = ix.search_torch(Qall[:1].contiguous(), ef, stats=True)
            t = float(np.median(ms[4:]))
        else:
            Q = Qall[:nq].contiguous()
            out = ix.search_torch(Q, ef, stats=True)
            for _ in range(5):
                ix.search_torch(Q, ef, out=out)
                ms.append(ix.last_search_ms())
            t = min(ms)
        torch.cuda.synchronize()
        c = 0
        for k in ("labels", "dists", "stats", "counts"):
            c = zlib.crc32(out[k].cpu().numpy().tobytes(), c)
        ref = c if ref is None else ref
        print(f"dim {dim} {metric} nq={nq:6d} <24,1>={s24}: kernel {t:8.4f} ms {nq / t * 1e3:10.0f} q/s identical={c == ref} [{ix.last_search_kernel()}]", flush=True)
