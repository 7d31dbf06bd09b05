"""Team form when a wave walks SEVERAL queries while its siblings help (last paragraph): a launch is
squeezed into a few blocks (HNSW_GPU_MAX_BLOCKS), so that every wave with queries takes many of them through the ticket
counter with its helpers attached — the schedule of a small launch whose other blocks start late, which the normal suites
never produce.  Neighbouring queries of one cluster follow each other, so that a package scored against the previous query
would be for elements the next walk pops too.  Every case is compared bit for bit with the one-wave form and printed BEFORE
it runs.  Run under `timeout`:

    timeout 600 python tests/experiments/team_second_walk_stress.py [rounds] [rows]
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

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
dev = torch.device("cuda", 0)
KEYS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_WPB", "HNSW_GPU_MAX_BLOCKS")      # (HNSW_GPU_TEAM_SPEC of the pending patch: as the caller set it)


def setenv(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def crc(out):
    c = 0
    for k in ("labels", "dists", "stats"):
        if k in out and out[k] is not None:
            c = zlib.crc32(out[k].cpu().numpy().tobytes(), c)
    return c


rng = np.random.default_rng(11)
bad = cases = 0
for dim, m, func in ((768, 16, pg.DIST_L2), (96, 16, pg.DIST_L2), (768, 32, pg.DIST_COSINE)):
    X = gmm_torch(n, dim, stream=0, device=dev)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, 200, 128, func), n)
    ix.append_torch(X)
    ix.link(0, n)
    torch.cuda.synchronize()
    for r in range(rounds):
        # queries = slightly moved copies of a few rows, one after the other: consecutive walks cross the same elements
        nq = int(rng.choice([8, 24, 64, 200]))
        base = X[torch.from_numpy(rng.integers(0, n, size=max(1, nq // 8))).to(dev)]
        Q = (base.repeat_interleave(8, dim=0)[:nq] + 0.01 * torch.randn(nq, dim, device=dev)).contiguous()
        ef = int(rng.choice([40, 128]))
        setenv({"HNSW_GPU_TEAM": "0"})
        ref = crc(ix.search_torch(Q, ef, stats=True))
        for blocks, wpb in (("1", "8"), ("2", "8"), ("3", "4"), ("1", "2")):
            print(f"dim {dim} m {m} func {func} round {r} nq {nq} ef {ef} blocks {blocks} wpb {wpb} ...", end="", flush=True)
            setenv({"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": wpb, "HNSW_GPU_MAX_BLOCKS": blocks})
            got = crc(ix.search_torch(Q, ef, stats=True))
            torch.cuda.synchronize()
            ok = got == ref
            bad += not ok
            cases += 1
            print(" same" if ok else " DIFFERENT", flush=True)
    del ix, X
print(f"cases: {cases} mismatches: {bad}")
sys.exit(1 if bad else 0)
