"""Random small configurations through the one-call inserts (csrc/device_insert.h) against the oracle's graph bytes — meant for the SIMT
emulator (PGEMB_GPU_LIB=tests/_build/libhnsw_gpu_simt.so python tests/experiments/insert_fuzz_emu.py [cases=24] [seed=1]); runs on a device too."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pg_embedding_amd as pg
import oracle
from pg_embedding_amd.datasets import gmm
from test_gpu_build import live_image

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for c in range(cases):
    func = int(rng.integers(0, 3))
    dim = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 100, 130]))
    m = int(rng.integers(1, 21))
    efc = int(rng.integers(1, 80))
    n = int(rng.integers(30, 140))
    X = gmm(n, dim, k=5, seed=100 + c)
    if n > 40:
        X[30:34] = X[10:14]                                # equal distances
    if func == 1:
        X += 0.01                                          # (no all-zero rows for cosine)
    labels = (np.arange(n, dtype=np.uint64) * 11 + 3)
    port = oracle.PortIndex(dim, m, efc, 64, func)
    port.add(X, labels)
    meta = pg.make_meta(dim, m, efc, 64, func)
    want = live_image(port.raw(), meta, n)
    ix = pg.GpuIndex.empty(meta, n)
    for i in range(n):
        cand = None
        if i > 0 and (i + c) % 2:
            ci, cd, pops, nev = ix.search_trace(X[i], efc, base=True)
            cand = (ci.astype(np.uint32), cd)
        ix.insert_one(X[i], int(labels[i]), candidates=cand)
    got = ix.export_flat().reshape(n, -1)
    diff = int((got != want).any(axis=1).sum())
    bad += diff != 0
    print(f"case {c}: func {func} dim {dim} m {m} efc {efc} n {n}: {diff} elements differ; paths {ix.insert_path_counts()}", flush=True)
    ix.close()
print("FAILED" if bad else "all exact")
sys.exit(1 if bad else 0)
