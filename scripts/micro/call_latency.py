"""End-to-end latency of ONE query per host-pointer call (hnsw_gpu_search_batch, what the drop-in hnsw_search does with an
attached mirror): wall clock around the call vs the kernel's own HIP-event time, polled zero-copy path vs copy path.
usage: call_latency.py [dim] [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 768
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, 16, 200, 128, pg.DIST_L2), n); ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize()
del X
Q = gmm_torch(256, dim, stream=1, device=dev).cpu().numpy()
ref = None
for mode, env in (("copies + stream wait", "1"), ("polled zero-copy", None)):
    if env: os.environ["HNSW_GPU_NO_POLL"] = env
    else: os.environ.pop("HNSW_GPU_NO_POLL", None)
    for i in range(16): ix.search(Q[i:i + 1], 128)
    wall, kern, outs = [], [], []
    for i in range(256):
        q = Q[i:i + 1]
        t0 = time.perf_counter(); lab, dst, cnt = ix.search(q, 128); t1 = time.perf_counter()
        wall.append((t1 - t0) * 1e3); kern.append(ix.last_search_ms()); outs.append((lab.copy(), dst.copy(), cnt.copy()))
    if ref is None: ref = outs
    same = all((a[0] == b[0]).all() and (a[1].view(np.uint32) == b[1].view(np.uint32)).all() and (a[2] == b[2]).all() for a, b in zip(ref, outs))
    print(f"dim {dim} rows {n} one query per call, {mode:22s}: call median {np.median(wall):.3f} ms mean {np.mean(wall):.3f}  "
          f"kernel median {np.median(kern):.3f} ms  overhead {np.median(wall) - np.median(kern):.3f} ms  identical={same}", flush=True)
for nq in (4, 16):
    for mode, env in (("copies + stream wait", "1"), ("polled zero-copy", None)):
        if env: os.environ["HNSW_GPU_NO_POLL"] = env
        else: os.environ.pop("HNSW_GPU_NO_POLL", None)
        wall = []
        for i in range(0, 256 - nq, nq):
            t0 = time.perf_counter(); ix.search(Q[i:i + nq], 128); wall.append((time.perf_counter() - t0) * 1e3)
        print(f"dim {dim} {nq} queries per call, {mode:22s}: call median {np.median(wall):.3f} ms", flush=True)
