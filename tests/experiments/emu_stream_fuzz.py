"""Differential fuzz of the stream path on the SIMT emulator: random index shapes, beams, rings, walking waves per block and wave-schedule
jitter; every answer that comes back through a stream must equal the oracle's.  Not part of the test tiers (minutes of CPU):
    python tests/experiments/emu_stream_fuzz.py [iterations] [seed]
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
import numpy as np                                         # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
import util as U                                           # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad_total = 0
for it in range(iters):
    dim = int(rng.choice([8, 24, 33, 64, 100, 128, 129, 200, 300, 520]))
    m = int(rng.choice([4, 8, 16]))
    func = int(rng.choice([pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN]))
    ef = int(rng.choice([1, 5, 16, 40, 64, 100, 128, 200, 256]))
    walkers = int(rng.integers(1, 9))
    ring = int(rng.choice([64, 128]))
    n = int(rng.integers(300, 1500))
    nq = int(rng.integers(30, 110))
    jitter = int(rng.choice([0, 0, 2, 4]))
    light = int(rng.choice([1, 1, 0]))
    os.environ["SIMT_EMU_CUS"] = str(int(rng.choice([2, 3])))
    os.environ["SIMT_EMU_JITTER"] = str(jitter)
    t0 = time.time()
    port, X = U.build_port(n, dim, m, 40, func, k=10, seed=1000 + it)
    Q = gmm(nq, dim, k=10, seed=2000 + it)
    want = port.search_many(Q, ef, nthreads=4)
    ix = U.mirror(port, func, efs=ef)
    ctx = pg.SearchContext(ix)
    pg.config_set("HNSW_GPU_STREAM_LIGHT", None if light else 0)
    cfg = {"it": it, "dim": dim, "m": m, "func": func, "ef": ef, "walkers": walkers, "ring": ring, "n": n, "nq": nq, "jitter": jitter, "light": light}
    try:
        st = pg.SearchStream(ctx, ef, ring=ring, walkers=walkers)
    except RuntimeError as e:
        print(json.dumps(dict(cfg, refused=str(e)[:120])), flush=True)
        ctx.close(); ix.close()
        continue
    bad = 0
    try:
        done, pending = 0, []
        while done < nq or pending:
            inflight = sum(len(sl) for _, sl in pending)
            if done < nq and inflight <= ring // 2:
                k = int(min(nq - done, rng.integers(1, ring // 2 - 1), ring - inflight - 1))
                pending.append((done, st.submit(Q[done:done + k])))
                done += k
                continue
            first, slots = pending.pop(0)
            lab, dst, cnt = st.wait(slots, timeout=600.0)
            for j in range(len(slots)):
                q = first + j
                same = (lab[j] == want["labels"][q]).all() and (U.bits(dst[j]) == U.bits(want["dists"][q])).all() and cnt[j] == want["counts"][q]
                bad += 0 if same else 1
    finally:
        pg.config_set("HNSW_GPU_STREAM_LIGHT", None)
        st.close()
    ctx.close(); ix.close()
    bad_total += bad
    print(json.dumps(dict(cfg, wrong=bad, seconds=round(time.time() - t0, 1))), flush=True)
print("TOTAL WRONG", bad_total)
sys.exit(1 if bad_total else 0)
