"""Where a serial insert's time goes: hnsw_gpu_index_insert_one (append + serial link + gather, one polled wait)
on an attached mirror, its pieces one by one, and a traced one-query walk with ef = efConstruction (what the validated cache of the
unmodified glue runs in front of every insert).   python tests/experiments/insert_latency.py [rows=20000] [dims=128] [serial] [--timeout S]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd._lib import check
from pg_embedding_amd.datasets import gmm

sync = torch.cuda.synchronize if torch.cuda.is_available() else (lambda: None)      # (the emulated library needs none)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m, efc, extra = 16, 64, 2000
X = gmm(n + extra, dim, k=100, seed=5)
meta = pg.make_meta(dim, m, efc, 64, pg.DIST_L2)
ix = pg.GpuIndex.empty(meta, n + extra + 16)
ix.append(X[:n])
serial_base = "serial" in sys.argv[3:]                     # the first n rows linked one by one (the reference's own graph: fuller lists) instead of in batches
ix.link(0, n, max_batch=1) if serial_base else ix.link(0, n)
if serial_base:
    print(f"(base graph of {n} rows built by serial inserts)", flush=True)
sync()
L, h = ix.L, ix._h
maxM = int(meta.maxM)
mine = (C.c_uint32 * (maxM + 1))()
others = (C.c_uint32 * (maxM * (maxM + 1)))()


def med(ts):
    return float(np.median(ts)) * 1e3


# 1. the one-call inserts: two launches built for latency (device_insert.h) against the general builder path (HNSW_GPU_INSERT_FUSED=0),
#    with the insert's own walk (insert_one) and with the walk's result handed over (insert_candidates: what the shim does)
nxt = n
for fused in ("1", "0"):
    os.environ["HNSW_GPU_INSERT_FUSED"] = fused
    what = "two launches (device_insert.h)" if fused == "1" else "general builder path        "
    ts = []
    for i in range(extra // 8):
        p = np.ascontiguousarray(X[nxt])
        t0 = time.perf_counter()
        check(L.hnsw_gpu_index_insert_one(h, p.ctypes.data, nxt, nxt, mine, others), "insert_one")
        ts.append(time.perf_counter() - t0)
        nxt += 1
    print(f"{n} x {dim} m={m} efc={efc}: hnsw_gpu_index_insert_one,        {what}: median {med(ts):.3f} ms  (p10 {np.percentile(ts, 10) * 1e3:.3f}, p90 {np.percentile(ts, 90) * 1e3:.3f})", flush=True)
    ts, tw = [], []
    for i in range(extra // 8):
        p = np.ascontiguousarray(X[nxt])
        t0 = time.perf_counter()
        ci, cd, pops, nev = ix.search_trace(p, efc, base=True)
        t1 = time.perf_counter()
        ci32 = np.ascontiguousarray(ci.astype(np.uint32)); cd32 = np.ascontiguousarray(cd, dtype=np.float32)
        t2 = time.perf_counter()
        check(L.hnsw_gpu_index_insert_candidates(h, p.ctypes.data, nxt, nxt, ci32.ctypes.data, cd32.ctypes.data, len(ci32), mine, others), "insert_candidates")
        ts.append(time.perf_counter() - t2); tw.append(t1 - t0)
        nxt += 1
    print(f"{n} x {dim} m={m} efc={efc}: hnsw_gpu_index_insert_candidates, {what}: median {med(ts):.3f} ms  (p10 {np.percentile(ts, 10) * 1e3:.3f}, p90 {np.percentile(ts, 90) * 1e3:.3f}) behind a traced walk of {med(tw):.3f} ms", flush=True)
os.environ.pop("HNSW_GPU_INSERT_FUSED")
paths = (C.c_uint64 * 2)()
L.hnsw_gpu_insert_path_counts(paths)
print(f"   inserts by path: two launches {paths[0]}, general {paths[1]}", flush=True)
# 2. the round-2 sequence: append (blocking copies) + link + get_link_lists
ts, ta, tl, tg = [], [], [], []
base = nxt
for i in range(extra // 2):
    p = np.ascontiguousarray(X[base + i])
    lab = np.asarray([base + i], np.uint64)
    t0 = time.perf_counter()
    check(L.hnsw_gpu_index_append(h, p.ctypes.data, lab.ctypes.data, 1), "append")
    t1 = time.perf_counter()
    check(L.hnsw_gpu_index_link(h, base + i, 1, 1, 0, None), "link")
    sync()
    t2 = time.perf_counter()
    check(L.hnsw_gpu_index_get_link_lists(h, base + i, mine, others), "get_link_lists")
    t3 = time.perf_counter()
    ts.append(t3 - t0); ta.append(t1 - t0); tl.append(t2 - t1); tg.append(t3 - t2)
print(f"   append + link + get_link_lists (three calls)      median {med(ts):.3f} ms = append {med(ta):.3f} + link (incl. sync) {med(tl):.3f} + gather {med(tg):.3f}", flush=True)
# 3. one traced walk with ef = efc in base mode (the validation walk of shim_cache.h), and the plain one-query search
ts, tk = [], []
for i in range(300):
    q = np.ascontiguousarray(X[(i * 37) % n])
    t0 = time.perf_counter()
    ix.search_trace(q, efc, base=True)
    ts.append(time.perf_counter() - t0)
    tk.append(ix.last_search_ms())
print(f"   traced one-query walk, ef = {efc} (hnsw_gpu_search_trace): median {med(ts):.3f} ms per call, kernel {float(np.median(tk)):.3f} ms  [{ix.last_search_kernel()}]", flush=True)
ts = []
Q = np.ascontiguousarray(X[:1])
for i in range(300):
    t0 = time.perf_counter()
    ix.search(np.ascontiguousarray(X[(i * 37) % n:(i * 37) % n + 1]), efc)
    ts.append(time.perf_counter() - t0)
print(f"   plain one-query hnsw_gpu_search_batch, ef = {efc}: median {med(ts):.3f} ms per call, kernel {ix.last_search_ms():.3f} ms", flush=True)
