"""Slice helpers A/B (HNSW_GPU_TEAM_SPEC = how many of a walk's helpers speculate; the others score slices of its many-row
hops): kernel time per launch size, every output CRC-compared with the one-wave form.
usage: exp_spec_ab.py <dim> <m> [metric] [--timeout S]     env: EXP_ROWS, EXP_NQS, EXP_SPECS"""
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
Qall = gmm_torch(160000, dim, stream=1, device=dev)
KEYS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_SPEC")


def crc(out):
    c = 0
    for k in ("labels", "dists", "stats", "counts"):
        c = zlib.crc32(out[k].cpu().numpy().tobytes(), c)
    return c


specs = os.environ.get("EXP_SPECS", "8,5,3,0").split(",")
for nq in [int(x) for x in os.environ.get("EXP_NQS", "1,16,256,1024,10000,40000").split(",")]:
    ref = None
    for name, env in [("one-wave", {"HNSW_GPU_TEAM": "0"})] + [(f"team spec {s}", {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_SPEC": s}) for s in specs] + [("default", {})]:
        for k in KEYS:
            os.environ.pop(k, None)
        os.environ.update(env)
        print(f"dim {dim} nq={nq:6d} {name:12s} ...", end="", flush=True)
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
        c = crc(out)
        if ref is None:
            ref = c
        print(f" kernel {t:8.4f} ms {nq / t * 1e3:10.0f} q/s slots {ix.last_search_slots():5d} identical={c == ref} [{ix.last_search_kernel()}]", flush=True)
print("health", ix.health())
