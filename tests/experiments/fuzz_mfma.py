"""Randomised parity sweep of the exhaustive scorer: the MFMA filter + canonical re-score (csrc/device_bf_mfma.h) against the all-canonical
scan, ids and distance bits, over random table sizes (around tile edges), dimensions (rows that end inside a K step), query counts (one to
several 128-query tiles), both block tiles, k, L2 / cosine, exact ties and duplicate rows, queries equal to rows.  Prints one line per case; exits non-zero on
the first mismatch.   usage: fuzz_mfma.py <cases> [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm


def one_case(rng, idx):
    func = int(rng.choice([pg.DIST_L2, pg.DIST_COSINE]))
    dim = int(rng.choice([1, 3, 4, 7, 30, 31, 32, 33, 36, 63, 64, 65, 100, 127, 128, 129, 200, 257, 384, 500, 768, 769, 1000, 1536, 1900]))
    n = int(rng.choice([4096, 4097, 4223, 4224, 5000, 8191, 8192, 8193, 12345, 20000, 33000, 70001]))
    if dim > 800:
        n = min(n, 20000)
    nq = int(rng.choice([1, 2, 63, 64, 127, 128, 129, 200, 256, 257, 500]))
    k = int(rng.choice([1, 2, 5, 10, 32, 100]))
    X = gmm(n, dim, k=int(rng.integers(1, 60)), sigma=float(rng.choice([0.05, 0.3, 1.0])), seed=2000 + idx)
    if rng.random() < 0.3:
        X = np.rint(X * 4).astype(np.float32)                  # many exact ties
    if rng.random() < 0.4:
        X[n // 2:n // 2 + n // 10] = X[:n // 10]                # duplicate rows: equal distances, the lower idx wins
    if func == pg.DIST_COSINE:
        X[(X * X).sum(axis=1) == 0] = 1.0                       # (zero vectors: NaN cosine distances are outside the parity contract)
    Q = gmm(nq, dim, k=20, seed=3000 + idx, stream=1)
    if rng.random() < 0.5:
        Q[: min(nq, 3)] = X[rng.integers(0, n, size=min(nq, 3))]   # zero distances
    if func == pg.DIST_COSINE:
        Q[(Q * Q).sum(axis=1) == 0] = 1.0
    ix = pg.GpuIndex.empty(pg.make_meta(dim, 4, 8, 8, func), n)
    ix.append(X)
    dq = torch.from_numpy(Q).cuda()
    i0, d0 = ix.bruteforce_torch(dq, k)
    tile = str(rng.choice(["128x128", "256x256"]))             # (the library picks 256 x 256 only for launches with thousands of tiles: forced here)
    pg._lib.gpu_lib().hnsw_gpu_config_set(b"HNSW_GPU_BF_BIG_MIN_BLOCKS", b"0" if tile == "128x128" else b"-1")
    i1, d1 = ix.bruteforce_torch(dq, k, mfma=True)
    torch.cuda.synchronize()
    ok = bool((i0 == i1).all().item()) and bool((d0.view(torch.int32) == d1.view(torch.int32)).all().item())
    ix.close()
    print(f"case {idx}: func {func} n {n} dim {dim} nq {nq} k {k} tile {tile}: {'ok' if ok else 'MISMATCH'}", flush=True)
    return ok


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    for i in range(cases):
        if not one_case(rng, seed * 100000 + i):
            sys.exit(1)
    print(f"{cases} cases, seed {seed}: all identical to the canonical scan")
