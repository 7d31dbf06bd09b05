"""Randomised parity at larger sizes: graphs built on the device (batched), searched on the device and
by the multi-threaded CPU oracle on the exported bytes; covers the hash-set spill, candidate-set
overflow handling, every shape of the row loader, ef up to 512 (LDS form) and vacuum flags."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import oracle, pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda", 0)
bad = 0
for case in range(int(sys.argv[1])):
    func = int(rng.integers(0, 3))
    dim = int(rng.choice([8, 48, 100, 128, 200, 256, 320, 384, 500, 512, 640, 768, 1024, 1536]))
    m = int(rng.choice([4, 8, 16, 32, 48]))
    n = int(rng.integers(20000, 150000 if dim <= 512 else 60000))
    efc = int(rng.choice([16, 64, 200]))
    ef = int(rng.choice([10, 64, 128, 200, 256, 300, 512]))
    k = int(rng.choice([1, 20, 300]))
    sigma = float(rng.choice([0.1, 0.3, 1.0]))
    X = gmm_torch(n, dim, k=k, sigma=sigma, seed=case + 17, device=dev)
    Q = gmm_torch(400, dim, k=k, sigma=sigma, seed=case + 17, stream=1, device=dev)
    if rng.random() < 0.3:
        X = torch.round(X * 3); Q = torch.round(Q * 3)
    if func == 1:
        X[(X * X).sum(1) == 0] = 1.0; Q[(Q * Q).sum(1) == 0] = 1.0
    meta = pg.make_meta(dim, m, efc, ef, func)
    ix = pg.GpuIndex.empty(meta, n); ix.append_torch(X.contiguous()); ix.link(0, n); torch.cuda.synchronize()
    ndel = int(rng.integers(0, n // 4))
    for i in rng.choice(n, min(ndel, 200), replace=False):
        ix.set_deleted(int(i))
    out = ix.search_torch(Q.contiguous(), ef, stats=True); torch.cuda.synchronize()
    port = oracle.PortIndex(dim, m, efc, ef, func, capacity=n); port.load_raw(ix.export_flat(), n)
    W = port.search_many(Q.cpu().numpy(), ef, nthreads=32)
    L = out["labels"].cpu().numpy().view(np.uint64); D = out["dists"].cpu().numpy(); Cn = out["counts"].cpu().numpy()
    st = out["stats"].cpu().numpy().astype(np.uint32)
    ok = (Cn == W["counts"]).all() and (st[:, 0] == W["evals"]).all() and (st[:, 1] == W["hops"]).all()
    for q in range(400):
        c = int(W["counts"][q])
        ok = ok and (L[q, :c] == W["labels"][q, :c]).all() and (D[q, :c].view(np.uint32) == W["dists"][q, :c].view(np.uint32)).all()
    print(f"case {case}: func={func} dim={dim} m={m} n={n} efc={efc} ef={ef} k={k} maxE={int(W['evals'].max())} {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += (not ok)
    ix.close(); del X, Q, port
print("ALL OK" if bad == 0 else f"{bad} FAILED")
sys.exit(1 if bad else 0)
