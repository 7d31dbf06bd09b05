"""Back-to-back launches of one batch on ONE stream: wall-clock per launch against the kernel's own time, and how long each call holds the host.
usage: exp_launch_gap.py <dim> <m> <metric> <sift 0|1> [nq] [reps]     env: EXP_ROWS; knobs through HNSW_GPU_* (PGEMB_ENV_SYNC)"""
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
for label, env in [("default", {}), ("HNSW_GPU_NARROW5=0", {"HNSW_GPU_NARROW5": "0"})]:
    for k in ("HNSW_GPU_LEAN", "HNSW_GPU_NARROW5", "HNSW_GPU_BLOCKS_PER_CU"):
        os.environ.pop(k, None)
    os.environ.update(env)
    out = ix.search_torch(Q, ef)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(dev)
    ctx = pg.SearchContext(ix)
    out2 = ix.search_torch(Q, ef)
    def mirror_side():
        with torch.cuda.stream(side):
            ix.search_torch(Q, ef, out=out)

    callers = (("mirror, null stream", lambda: ix.search_torch(Q, ef, out=out), lambda: ix.last_search_ms(0)),
               ("mirror, side stream", mirror_side, lambda: ix.last_search_ms(0)),
               ("context, null stream", lambda: ctx.search_torch(Q, ef, out2), lambda: float("nan")),
               ("context, side stream", lambda: ctx.search_torch(Q, ef, out2, side), lambda: float("nan")))
    for who, call, last_ms in callers:
        call(); torch.cuda.synchronize()
        host = []
        t0 = time.perf_counter()
        for _ in range(reps):
            t1 = time.perf_counter()
            call()
            host.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3 / reps
        kms = [last_ms()]
        print(f"{label:26s} {who:22s}: wall {wall:.3f} ms per launch, last kernel {sorted(kms)[len(kms) // 2]:.3f} ms, host time per call median "
              f"{sorted(host)[len(host) // 2]:.3f} max {max(host):.3f} ms, slots {ix.last_search_slots()} [{ix.last_search_kernel()}]", flush=True)
    ctx.close()
