"""The slow state of the narrow-row launch, per WORKSPACE: many search contexts on one mirror in one process (each owns a visited-set workspace
and its own abort word, 64 bytes of pinned host memory), every one timed with the abort word read at the top of EVERY query (rounds 1-5) and of
every 16th (round 6).  A slow context that becomes fast by asking the host 16 times less often is the root cause shown directly.
usage: exp_slow_state_many.py [contexts=12]     (in sub-processes: without / with a bench process's history, then with the process
       bound to the CPUs of each NUMA node of the host in turn: where the pinned page of a workspace's abort word sits follows the thread that asks for it)"""
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "child" not in sys.argv:
    n = sys.argv[1] if len(sys.argv) > 1 else "12"
    for hist in ("0", "1"):
        print(f"=== history={hist}", flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", hist, n], timeout=900)
    import glob
    allowed = os.sched_getaffinity(0)
    for node in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        cpus = set()
        for part in open(node + "/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus |= set(range(int(a), int(b or a) + 1))
        cpus &= allowed
        print(f"=== history=1, process bound to {os.path.basename(node)} ({len(cpus)} of its CPUs allowed)", flush=True)
        if cpus:
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", "1", n], timeout=900, env=dict(os.environ, EXP_CPUS=",".join(map(str, sorted(cpus)))))
    sys.exit(0)

if os.environ.get("EXP_CPUS"):
    os.sched_setaffinity(0, {int(c) for c in os.environ["EXP_CPUS"].split(",")})

from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
import bench

hist, nctx = sys.argv[2] == "1", int(sys.argv[3])
sys.argv = sys.argv[:1]
dev = torch.device("cuda", 0)
args = bench.parse()
args.n, args.efc, args.ef, args.max_batch, args.ratio = 1_000_000, 200, 128, 0, 0
if hist:
    from pg_embedding_amd.datasets import gmm_torch
    Xw = gmm_torch(200_000, 768, device=dev)
    iw = pg.GpuIndex.empty(pg.make_meta(768, 16, 200, 128, pg.DIST_L2), 200_000)
    iw.append_torch(Xw); iw.link(0, 200_000); torch.cuda.synchronize()
    Qw = gmm_torch(10_000, 768, stream=1, device=dev)
    ow = iw.search_torch(Qw, 128); torch.cuda.synchronize()
    bench.two_streams(iw, Qw, 128, ow["labels"], dev)
    iw.close(); del Xw, Qw, ow
    torch.cuda.empty_cache()
ix, Q = bench.build_side_config(args, bench.side_cases(dev)[0], dev, 0)
out = ix.search_torch(Q, 128, stats=True)
torch.cuda.synchronize()
st = out["stats"].cpu().numpy().astype(np.int64)
byt = float(bench.alg_bytes(st, out["counts"].cpu().numpy().astype(np.int64), 128, 16).sum())
ctxs = [pg.SearchContext(ix) for _ in range(nctx)]
o2 = ix.search_torch(Q, 128)
s = torch.cuda.current_stream(dev)


def med(fn, ms_of, reps=5):
    v = []
    for _ in range(reps):
        fn(); v.append(ms_of())
    return float(np.median(v[1:]))


rows = []
for log2 in (0, 4):
    pg.config_set("HNSW_GPU_ABORT_POLL_LOG2", log2)
    rows.append([med(lambda: ix.search_torch(Q, 128, out=out), ix.last_search_ms)] + [med(lambda c=c: c.search_torch(Q, 128, o2, s), c.last_search_ms) for c in ctxs])
pg.config_set("HNSW_GPU_ABORT_POLL_LOG2", None)
print("  workspace:                      default " + " ".join(f"ctx{i:<3d}" for i in range(nctx)))
print("  every query  (rounds 1-5), ms:  " + " ".join(f"{v:6.2f}" for v in rows[0]))
print("  every 16th   (round 6),    ms:  " + " ".join(f"{v:6.2f}" for v in rows[1]))
slow = [i for i, v in enumerate(rows[0]) if v > 1.15 * min(rows[0])]
print(f"  slow workspaces when every query asks the host: {len(slow)} of {nctx + 1}" + (f" (x{max(rows[0]) / min(rows[0]):.2f}); the same workspaces asking every 16th query: "
      f"{[round(rows[1][i], 2) for i in slow]} ms against {min(rows[1]):.2f} for the fastest" if slow else ""), flush=True)
