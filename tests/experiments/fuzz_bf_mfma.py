"""Randomised check of the MFMA exhaustive scorer against the canonical scan (bit-exact ids + distances)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
for case in range(int(sys.argv[1])):
    func = int(rng.integers(0, 2))
    dim = int(rng.choice([1, 3, 17, 31, 32, 33, 64, 100, 127, 200, 255, 256, 257, 500, 768, 1000, 1536]))
    n = int(rng.integers(4096, 60000))
    nq = int(rng.integers(1, 700))
    k = int(rng.choice([1, 2, 10, 32, 100]))
    X = gmm(n, dim, k=int(rng.integers(1, 50)), sigma=float(rng.choice([0.01, 0.3, 2.0])), seed=case)
    if rng.random() < 0.4:
        X = np.rint(X * 3).astype(np.float32)
    if func == 1:
        X[(X * X).sum(axis=1) == 0] = 1.0
    Q = gmm(nq, dim, k=5, seed=case, stream=1)
    if rng.random() < 0.3:
        Q[: max(1, nq // 3)] = X[: max(1, nq // 3)]
    if func == 1:
        Q[(Q * Q).sum(axis=1) == 0] = 1.0
    ix = pg.GpuIndex.empty(pg.make_meta(min(dim, 1900), 2, 4, 4, func), n)
    ix.append(X)
    dq = torch.from_numpy(Q).cuda()
    i0, d0 = ix.bruteforce_torch(dq, k)
    i1, d1 = ix.bruteforce_torch(dq, k, mfma=True)
    torch.cuda.synchronize()
    ok = bool((i0 == i1).all()) and bool((d0.view(torch.int32) == d1.view(torch.int32)).all())
    print(f"case {case}: func={func} dim={dim} n={n} nq={nq} k={k} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
    ix.close()
print("ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
