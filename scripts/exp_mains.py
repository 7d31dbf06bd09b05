"""Walking waves per block of a team launch (HNSW_GPU_TEAM_MAINS): fewer walks at a time, each with helpers from the start.
usage: exp_mains.py <dim> <m> [metric] [--timeout S]     env: EXP_NQS, EXP_MAINS"""
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
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
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


def crc(out):
    c = 0
    for k in ("labels", "dists", "stats", "counts"):
        c = zlib.crc32(out[k].cpu().numpy().tobytes(), c)
    return c


for nq in [int(x) for x in os.environ.get("EXP_NQS", "1024,2560,5000,10000,20000").split(",")]:
    ref = None
    for mains in ["8"] + os.environ.get("EXP_MAINS", "7,6,5,4").split(","):
        os.environ["HNSW_GPU_TEAM"] = "1"
        os.environ["HNSW_GPU_TEAM_MAINS"] = mains
        Q = Qall[:nq].contiguous()
        out = ix.search_torch(Q, ef, stats=True)
        ms = []
        for _ in range(5):
            ix.search_torch(Q, ef, out=out)
            ms.append(ix.last_search_ms())
        torch.cuda.synchronize()
        c = crc(out)
        ref = c if ref is None else ref
        t = min(ms)
        print(f"dim {dim} nq={nq:6d} walking waves per block {mains}: kernel {t:8.4f} ms {nq / t * 1e3:10.0f} q/s slots {ix.last_search_slots():5d} identical={c == ref}", flush=True)
