"""Differential fuzz of the one-call serial insert on the SIMT emulator: random metrics, dimensions, m, efConstruction, sizes, duplicate rows
(ties by element number) and wave-schedule jitter; after every run the mirror's element images must equal the oracle's (the reference's
insert order, hnswalg.cpp:117-232) byte for byte.  Not part of the test tiers:  python tests/experiments/emu_insert_fuzz.py [iterations] [seed]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu                                           # noqa: E402

os.environ["PGEMB_GPU_LIB"] = build_emu.build()
os.environ["PGEMB_ENV_SYNC"] = "1"
import numpy as np                                         # noqa: E402
import oracle                                              # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402
from test_gpu_build import live_image                      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
bad_total = 0
for it in range(iters):
    func = int(rng.choice([pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN]))
    dim = int(rng.choice([3, 6, 12, 20, 33, 64, 130]))
    m = int(rng.choice([1, 2, 4, 8, 16, 40]))
    efc = int(rng.choice([4, 16, 40, 64, 100, 600]))
    n = int(rng.integers(40, 160))
    fused = "1" if rng.random() < 0.8 else "0"
    mode = "candidates" if rng.random() < 0.4 else "one"
    os.environ["HNSW_GPU_INSERT_FUSED"] = fused
    os.environ["SIMT_EMU_JITTER"] = str(int(rng.choice([0, 0, 3])))
    t0 = time.time()
    X = gmm(n, dim, k=6, seed=5000 + it)
    ndup = int(rng.integers(0, 6))
    for _ in range(ndup):
        a, b = rng.integers(0, n, 2)
        X[a] = X[b]
    labels = np.arange(n, dtype=np.uint64) * 3 + 7
    port = oracle.PortIndex(dim, m, efc, 64, func)
    port.add(X, labels)
    meta = pg.make_meta(dim, m, efc, 64, func)
    maxM = int(meta.maxM)
    want = live_image(port.raw(), meta, n)
    ix = pg.GpuIndex.empty(meta, n)
    mine = (C.c_uint32 * (maxM + 1))()
    others = (C.c_uint32 * (maxM * (maxM + 1)))()
    rc = 0
    for i in range(n):
        p = np.ascontiguousarray(X[i])
        if mode == "candidates" and i > 0:
            ci, cd, pops, nev = ix.search_trace(p, efc, base=True)
            ci32 = np.ascontiguousarray(ci.astype(np.uint32)); cd32 = np.ascontiguousarray(cd, dtype=np.float32)
            rc = ix.L.hnsw_gpu_index_insert_candidates(ix._h, p.ctypes.data, int(labels[i]), i, ci32.ctypes.data, cd32.ctypes.data, len(ci32), mine, others)
        else:
            rc = ix.L.hnsw_gpu_index_insert_one(ix._h, p.ctypes.data, int(labels[i]), i, mine, others)
        if rc != 0:
            break
    got = ix.export_flat().reshape(-1, want.shape[1]) if rc == 0 else None
    bad = 1 if rc != 0 else int((got != want).any(axis=1).sum())
    bad_total += bad
    print(json.dumps({"it": it, "func": func, "dim": dim, "m": m, "efc": efc, "n": n, "dups": ndup, "mode": mode, "fused": fused,
                      "jitter": os.environ["SIMT_EMU_JITTER"], "rc": rc, "elements_that_differ": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
    ix.close()
print("TOTAL WRONG", bad_total)
sys.exit(1 if bad_total else 0)
