"""Narrow rows (BASELINE config C2, 1M x 128 L2): waves per block of the one-wave launch (HNSW_GPU_NARROW_WPB = 4 / 2 / 1) — ONE launch of
40 000 queries at a time, and launches back to back on two search contexts / streams (does the next launch fill the drain of the
previous one when a finished wave frees its LDS and slot at once?).  VERDICT r5 "next" #1, step 2: the cheaper experiment.
usage: exp_narrow_wpb.py [nq=40000] [reps=12]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900)
import numpy as np
import torch
import pg_embedding_amd as pg
import bench

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda", 0)
sys.argv = sys.argv[:1]                                 # (bench.parse() reads the command line)
args = bench.parse()
args.n, args.efc, args.ef, args.max_batch, args.ratio = 1_000_000, 200, 128, 0, 0
case = list(bench.side_cases(dev)[0])
case[5] = nq
ix, Q = bench.build_side_config(args, tuple(case), dev, 0)
out = ix.search_torch(Q, 128, stats=True)
torch.cuda.synchronize()
st = out["stats"].cpu().numpy().astype(np.int64)
byt = float(bench.alg_bytes(st, out["counts"].cpu().numpy().astype(np.int64), 128, 16).sum())
want = out["labels"].clone()
print(f"1M x 128 sift-like, m=16, ef=128, {nq} queries/launch: E_q {st[:, 0].mean():.1f} H_q {st[:, 1].mean():.1f}, {byt / 1e9:.2f} GB algorithmic per launch; placement {ix.placement()['aligned_2MiB']}", flush=True)
for rnd in range(2):                                    # (twice: a process-history effect would show between the rounds)
    for wpb in (4, 2, 1):
        pg.config_set("HNSW_GPU_NARROW_WPB", wpb)
        ms = []
        for _ in range(reps + 1):
            ix.search_torch(Q, 128, out=out)
            ms.append(ix.last_search_ms())
        ms = ms[1:]
        same = bool((out["labels"] == want).all().item())
        ts = bench.two_streams(ix, Q, 128, want, dev, reps)
        print(f"round {rnd} waves/block {wpb}: slots {ix.last_search_slots()} one launch at a time min/median/max {min(ms):.3f}/{float(np.median(ms)):.3f}/{max(ms):.3f} ms "
              f"= {byt / float(np.median(ms)) / 1e6 / 8000:.3f} of 8 TB/s (clock {ix.last_search_clock_mhz():.0f} MHz), identical {same}; two streams "
              f"{[round(byt * q / nq / 1e9 / 8000, 3) for q in ts['queries_per_s_of_each_round']]} of 8 TB/s, identical {ts['results_identical']}  [{ix.last_search_kernel()}]", flush=True)
pg.config_set("HNSW_GPU_NARROW_WPB", None)
