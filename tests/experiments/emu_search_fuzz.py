"""Differential fuzz of the search launches on the SIMT emulator: random index shapes, metrics, beams, batch sizes, kernel-form knobs and
wave-schedule jitter; labels, distance bits and counts must equal the oracle's.  Not part of the test tiers (minutes of CPU):
    python tests/experiments/emu_search_fuzz.py [iterations] [seed]
"""
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
os.environ["PGEMB_ENV_SYNC"] = "1"                         # kernel-form knobs follow os.environ (pg_embedding_amd/_lib.py)
import numpy as np                                         # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
import util as U                                           # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
KNOBS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_WPB", "HNSW_GPU_NARROW5", "HNSW_GPU_LEAN", "HNSW_GPU_HASH_ENTRIES", "HNSW_GPU_BEAM", "HNSW_GPU_MAX_BLOCKS",
         "HNSW_GPU_TEAM_SPEC")
bad_total = 0
for it in range(iters):
    dim = int(rng.choice([3, 8, 17, 32, 64, 96, 128, 129, 200, 256, 300, 520, 768]))
    m = int(rng.choice([2, 4, 8, 16, 24]))
    func = int(rng.choice([pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN]))
    ef = int(rng.choice([1, 3, 10, 33, 64, 100, 128, 200, 256, 300, 600]))
    n = int(rng.integers(50, 1400))
    nq = int(rng.integers(1, 40))
    env = {}
    if rng.random() < 0.6: env["HNSW_GPU_TEAM"] = str(int(rng.integers(0, 2)))
    if rng.random() < 0.3: env["HNSW_GPU_TEAM_WPB"] = str(int(rng.choice([2, 4, 8])))
    if rng.random() < 0.3: env["HNSW_GPU_NARROW5"] = "0"
    if rng.random() < 0.3: env["HNSW_GPU_LEAN"] = "0"
    if rng.random() < 0.3: env["HNSW_GPU_HASH_ENTRIES"] = str(int(rng.choice([0, 256, 512])))
    if rng.random() < 0.15: env["HNSW_GPU_BEAM"] = "0"
    if rng.random() < 0.3: env["HNSW_GPU_MAX_BLOCKS"] = str(int(rng.choice([1, 2])))
    if rng.random() < 0.3: env["HNSW_GPU_TEAM_SPEC"] = str(int(rng.choice([0, 1, 2, 8])))
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["SIMT_EMU_CUS"] = str(int(rng.choice([1, 2, 4, 64])))
    os.environ["SIMT_EMU_JITTER"] = str(int(rng.choice([0, 0, 2, 4])))
    pg.sync_env()
    t0 = time.time()
    port, X = U.build_port(n, dim, m, 30, func, k=8, seed=3000 + it)
    Q = gmm(nq, dim, k=8, seed=4000 + it)
    want = port.search_many(Q, ef, nthreads=4)
    ix = U.mirror(port, func, efs=ef)
    try:
        lab, dst, cnt = ix.search(Q, ef)
    except RuntimeError as e:
        print(json.dumps({"it": it, "dim": dim, "ef": ef, "env": env, "error": str(e)[:200]}), flush=True)
        bad_total += 1
        ix.close()
        continue
    bad = 0
    for q in range(nq):
        c = int(want["counts"][q])                            # (rows are padded behind the count, each side in its own way)
        same = cnt[q] == c and (lab[q, :c] == want["labels"][q, :c]).all() and (U.bits(dst[q, :c]) == U.bits(want["dists"][q, :c])).all()
        bad += 0 if same else 1
    bad_total += bad
    print(json.dumps({"it": it, "dim": dim, "m": m, "func": func, "ef": ef, "n": n, "nq": nq, "env": env, "cus": os.environ["SIMT_EMU_CUS"],
                      "jitter": os.environ["SIMT_EMU_JITTER"], "kernel": ix.last_search_kernel(), "wrong": bad, "seconds": round(time.time() - t0, 1)}), flush=True)
    ix.close()
print("TOTAL WRONG", bad_total)
sys.exit(1 if bad_total else 0)
