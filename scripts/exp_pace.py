"""What a launch of one walk per walking wave could gain from serving the walks with the most hops left first: ORACLE pacing.
The experiment build (build.py variant pace HNSW_PACE) takes every query's hop count from an earlier launch of the same batch and holds
a walk that is ahead of the schedule "hop k of h at k / h of T" back (device_search.h, HNSW_PACE), so that every walk ends at about T:
short walks leave the memory system to the long ones, whose hops then cost what a lone walk's cost.  Sweeps T; results must not change.
usage: PGEMB_GPU_LIB=pg_embedding_amd/lib/variants/libhnsw_gpu_pace.so exp_pace.py <dim> <m> <metric l2|cosine> [nq=1024]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd import _lib
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, 128
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, stream=0, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Q = gmm_torch(nq, dim, stream=1, device=dev)
ref = ix.search_torch(Q, ef, stats=True)
torch.cuda.synchronize()
st = ref["stats"].cpu().numpy().astype(np.int64)
cnt = ref["counts"].cpu().numpy().astype(np.int64)
byt = float((st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum())
hops_dev = ref["stats"][:, 1].contiguous().to(torch.int32)
L = _lib.gpu_lib()
L.hnsw_gpu_experiment_pace.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
out = ix.search_torch(Q, ef, stats=True)


def timed(reps=7):
    ms = []
    for _ in range(reps):
        ix.search_torch(Q, ef, out=out)
        ms.append(ix.last_search_ms())
    same = bool(torch.equal(out["labels"], ref["labels"]) and torch.equal(out["dists"], ref["dists"]))
    return float(np.median(ms[1:])), float(np.min(ms)), same


print(f"dim {dim} m {m} {metric} nq {nq}: hops mean {st[:, 1].mean():.1f} max {st[:, 1].max()}; {byt / 1e9:.3f} GB algorithmic per launch", flush=True)
med, best, same = timed()
print(f"  no pacing:           launch median {med:.3f} ms (min {best:.3f}) = {byt / (med * 1e-3) / 8e12:.3f} of 8 TB/s, identical {same}  [{ix.last_search_kernel()}]", flush=True)
base = med
for frac in (0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9, 1.0):
    T_us = base * 1e3 * frac
    assert L.hnsw_gpu_experiment_pace(ix._h, hops_dev.data_ptr(), int(T_us * 100)) == 0
    med, best, same = timed()
    print(f"  schedule T = {T_us / 1e3:.3f} ms: launch median {med:.3f} ms (min {best:.3f}) = {byt / (med * 1e-3) / 8e12:.3f} of 8 TB/s, identical {same}", flush=True)
assert L.hnsw_gpu_experiment_pace(ix._h, None, 0) == 0
med, best, same = timed()
print(f"  no pacing (again):   launch median {med:.3f} ms (min {best:.3f}), identical {same}")
