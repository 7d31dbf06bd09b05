"""The slow state of the narrow-row launch (profiles/r5af_*, r6j): which launches of a process are slow — by workspace (the mirror's default one /
a search context's own) and by stream (torch's default stream / a stream of its own) — with and without the history of a bench process
(two contexts + two streams used on another index first) and with GPU_MAX_HW_QUEUES unset / 16.
usage: exp_slow_state.py            (runs itself in sub-processes, one per environment)
       exp_slow_state.py child <history 0|1>"""
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) < 2 or sys.argv[1] != "child":
    for hwq in (None, "4"):
        for hist in ("0", "1"):
            env = dict(os.environ)
            env.pop("GPU_MAX_HW_QUEUES", None)
            if hwq:
                env["GPU_MAX_HW_QUEUES"] = hwq
            print(f"=== GPU_MAX_HW_QUEUES={hwq or 'unset'} history={hist}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", hist], env=env, timeout=600)
    sys.exit(0)

from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
import bench

hist = sys.argv[2] == "1"
sys.argv = sys.argv[:1]
dev = torch.device("cuda", 0)
args = bench.parse()
args.n, args.efc, args.ef, args.max_batch, args.ratio = 1_000_000, 200, 128, 0, 0
if hist:
    # what a bench process has done before it reaches the side configs: a wide-row index searched on the default stream and through two
    # contexts on two streams
    from pg_embedding_amd.datasets import gmm_torch
    Xw = gmm_torch(200_000, 768, device=dev)
    iw = pg.GpuIndex.empty(pg.make_meta(768, 16, 200, 128, pg.DIST_L2), 200_000)
    iw.append_torch(Xw); iw.link(0, 200_000); torch.cuda.synchronize()
    Qw = gmm_torch(10_000, 768, stream=1, device=dev)
    ow = iw.search_torch(Qw, 128); torch.cuda.synchronize()
    bench.two_streams(iw, Qw, 128, ow["labels"], dev)
    iw.close(); del Xw, Qw, ow
    torch.cuda.empty_cache()
case = bench.side_cases(dev)[0]
ix, Q = bench.build_side_config(args, case, dev, 0)
out = ix.search_torch(Q, 128, stats=True)
torch.cuda.synchronize()
st = out["stats"].cpu().numpy().astype(np.int64)
byt = float(bench.alg_bytes(st, out["counts"].cpu().numpy().astype(np.int64), 128, 16).sum())
want = out["labels"].clone()


def timed(label, fn, ms_of):
    ms = []
    for _ in range(9):
        fn()
        ms.append(ms_of())
    ms = ms[1:]
    print(f"  {label:58s} min/median/max {min(ms):.3f}/{float(np.median(ms)):.3f}/{max(ms):.3f} ms = {byt / float(np.median(ms)) / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


s1 = torch.cuda.Stream(dev)
ctx = pg.SearchContext(ix)
o2 = ix.search_torch(Q, 128)


def on_s1():
    with torch.cuda.stream(s1):
        ix.search_torch(Q, 128, out=out)


# the abort word of a workspace lives in pinned host memory; rounds 1-5 every wave read it at the top of EVERY query (HNSW_GPU_ABORT_POLL_LOG2 = 0),
# round 6 reads it at the top of every 16th (= 4, the default)
for log2 in (0, 4, 0, 4):
    pg.config_set("HNSW_GPU_ABORT_POLL_LOG2", log2)
    print(f" abort word read at the top of every {1 << log2}{'th' if log2 else ''} query:", flush=True)
    timed("default workspace, torch's default stream", lambda: ix.search_torch(Q, 128, out=out), ix.last_search_ms)
    timed("default workspace, a stream of its own", on_s1, ix.last_search_ms)
    timed("a context's workspace, torch's default stream", lambda: ctx.search_torch(Q, 128, o2, torch.cuda.current_stream(dev)), ctx.last_search_ms)
    timed("a context's workspace, a stream of its own", lambda: ctx.search_torch(Q, 128, o2, s1), ctx.last_search_ms)
pg.config_set("HNSW_GPU_ABORT_POLL_LOG2", None)
print(f"  clock {ix.last_search_clock_mhz():.0f} MHz, results identical {bool((out['labels'] == want).all().item()) and bool((o2['labels'] == want).all().item())}, "
      f"placement {[(k, v[0]) for k, v in ix.placement().items() if isinstance(v, tuple)][:3]}", flush=True)
