"""Does what a process did BEFORE it built a narrow-row index change how fast that index is searched?  (bench run r5ac: C2 at 8.25 ms per
40 000-query launch, all 12 launches, against 5.8-5.9 ms in every other run of the same kernel — the legs in front of it differed.)
usage: exp_alloc_state.py <history>   history = fresh | wide_first | wide_first_empty_cache | torch_hog"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

hist = sys.argv[1]
dev = torch.device("cuda", 0)
ef = 128


def sift_rows(cnt, dim, stream):
    return torch.clamp(torch.round(40.0 + 35.0 * gmm_torch(cnt, dim, stream=stream, device=dev)), 0, 218)


def build(dim, m, n, rows):
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, 200, ef, pg.DIST_L2), n)
    ix.append_torch(rows); ix.link(0, n); torch.cuda.synchronize()
    return ix


def timed(ix, Q, reps=8):
    out = ix.search_torch(Q, ef, stats=True)
    ms = []
    for _ in range(reps):
        ix.search_torch(Q, ef, out=out); torch.cuda.synchronize(); ms.append(ix.last_search_ms())
    return float(np.median(ms)), float(min(ms)), float(max(ms))


if hist.startswith("wide_first"):
    X = gmm_torch(1_000_000, 768, device=dev)
    ixw = build(768, 16, 1_000_000, X); del X
    Qw = gmm_torch(40000, 768, stream=1, device=dev)
    print("wide index first: 40 000 queries %.3f ms (min %.3f max %.3f)" % timed(ixw, Qw, 4), flush=True)
    ixw.close(); del ixw, Qw
    if hist == "wide_first_empty_cache":
        torch.cuda.empty_cache()
elif hist == "torch_hog":
    hog = [torch.empty(1 << 30, dtype=torch.uint8, device=dev) for _ in range(24)]      # 24 GB through torch's allocator, then dropped (cached, not freed)
    del hog
X = sift_rows(1_000_000, 128, 0)
ix = build(128, 16, 1_000_000, X); del X
Q = sift_rows(40000, 128, 1)
print("history %-24s: 1M x 128, 40 000 queries per launch: kernel median %.3f ms (min %.3f max %.3f)  torch reserved %.1f GB  [%s]"
      % ((hist,) + timed(ix, Q) + (torch.cuda.memory_reserved() / 2**30, ix.last_search_kernel())), flush=True)
