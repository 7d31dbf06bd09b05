"""Launches of one batch back to back on 1 / 2 / 3 search contexts and streams: does the next launch fill the drain of the previous one?
usage: exp_two_streams.py <dim> <m> <metric l2|cosine> <sift 0|1> [nq] [reps]      env: GPU_MAX_HW_QUEUES (set before HIP starts), EXP_ROWS"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric, sift = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
nq = int(sys.argv[5]) if len(sys.argv) > 5 else 40000
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 12
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, 128
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)


def rows(cnt, stream):
    X = gmm_torch(cnt, dim, stream=stream, device=dev)
    return torch.clamp(torch.round(40.0 + 35.0 * X), 0, 218) if sift else X


X = rows(n, 0)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Q = rows(nq, 1)
out = ix.search_torch(Q, ef, stats=True)
torch.cuda.synchronize()
st = out["stats"].cpu().numpy().astype(np.int64)
cnt = out["counts"].cpu().numpy().astype(np.int64)
byt = float((st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum())
ms = []
for _ in range(reps):
    ix.search_torch(Q, ef, out=out)
    ms.append(ix.last_search_ms())
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} dim {dim} nq {nq}: one stream, kernel min/median/max "
      f"{min(ms):.3f}/{sorted(ms)[len(ms) // 2]:.3f}/{max(ms):.3f} ms = {byt / sorted(ms)[len(ms) // 2] / 1e6 / 8000:.3f} of 8 TB/s [{ix.last_search_kernel()}]", flush=True)
want = out["labels"].clone()
# what a pause does to the launches after it: the same batch, 8 launches behind 100 ms of idle
torch.cuda.synchronize()
time.sleep(0.1)
ms = []
for _ in range(8):
    ix.search_torch(Q, ef, out=out)
    ms.append(ix.last_search_ms())
print("   kernel ms of 8 launches behind 100 ms of idle: " + " ".join(f"{x:.3f}" for x in ms), flush=True)
WARM = int(os.environ.get("EXP_WARM", "1"))
for ns in (1, 2, 3, 4):
    ctxs = [pg.SearchContext(ix) for _ in range(ns)]
    streams = [torch.cuda.Stream(dev) for _ in range(ns)]
    outs = [ix.search_torch(Q, ef) for _ in range(ns)]
    torch.cuda.synchronize()
    for i in range(ns * WARM):
        ctxs[i % ns].search_torch(Q, ef, outs[i % ns], streams[i % ns])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        ctxs[i % ns].search_torch(Q, ef, outs[i % ns], streams[i % ns])
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    ok = all(bool((o["labels"] == want).all().item()) for o in outs)
    print(f"   {ns} contexts / streams, {WARM} warm-up launches each: {reps} launches in {t * 1e3:.2f} ms = {t * 1e3 / reps:.3f} ms per launch = {nq * reps / t:.0f} q/s = "
          f"{byt * reps / t / 1e9 / 8000:.3f} of 8 TB/s, identical={ok}", flush=True)
    for c in ctxs:
        c.close()
