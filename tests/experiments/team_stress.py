"""Stress of the team form with slice helpers (HNSW_GPU_TEAM_SPEC=5): thousands of small launches of every shape that gives
a block helpers, each compared bit for bit with the one-wave form.  Every case prints BEFORE it runs (flush), so a launch
that never returns is the last line of the log.  Run under `timeout`:

    timeout 600 python tests/experiments/team_stress.py [rounds] [rows]
"""
import os
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
dev = torch.device("cuda", 0)
KEYS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_SPEC", "HNSW_GPU_TEAM_WPB")


def setenv(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def crc(out):
    c = 0
    for k in ("labels", "dists", "stats", "counts"):
        if k in out and out[k] is not None:
            c = zlib.crc32(out[k].cpu().numpy().tobytes(), c)
    return c


rng = np.random.default_rng(7)
bad = 0
for dim, m, func in ((768, 16, pg.DIST_L2), (768, 32, pg.DIST_COSINE), (96, 16, pg.DIST_L2), (1536, 32, pg.DIST_COSINE)):
    ef = 128
    X = gmm_torch(n, dim, stream=0, device=dev)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, 200, ef, func), n)
    ix.append_torch(X)
    ix.link(0, n)
    torch.cuda.synchronize()
    Qall = gmm_torch(4096, dim, stream=1, device=dev)
    for r in range(rounds):
        nq = int(rng.choice([1, 1, 1, 2, 3, 5, 8, 13, 17, 64, 200, 256, 257, 700, 2100]))
        o = int(rng.integers(0, 4096 - nq + 1))
        efq = int(rng.choice([16, 40, 128, 200]))
        Q = Qall[o:o + nq].contiguous()
        setenv({"HNSW_GPU_TEAM": "0"})
        ref = crc(ix.search_torch(Q, efq, stats=True))
        for spec, wpb in (("5", "8"), ("2", "8"), ("0", "8"), ("1", "4"), ("8", "8")):
            print(f"dim {dim} m {m} func {func} round {r} nq {nq} ef {efq} spec {spec} wpb {wpb} ...", end="", flush=True)
            setenv({"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_SPEC": spec, "HNSW_GPU_TEAM_WPB": wpb})
            got = crc(ix.search_torch(Q, efq, stats=True))
            torch.cuda.synchronize()
            ok = got == ref
            bad += not ok
            print(" same" if ok else " DIFFERENT", flush=True)
    del ix, X
print("mismatches:", bad)
sys.exit(1 if bad else 0)
