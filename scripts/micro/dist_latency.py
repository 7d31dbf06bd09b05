"""Latency of ONE distance through the drop-in symbol hnsw_dist_func() (what the SQL operators <-> <=> <~>
call per row, embedding.c:1037): in-process (libembedding_gpu.so) and through hnsw_gpu_server (DIST request)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np                                            # noqa: E402
from pg_embedding_amd._lib import shim_lib                    # noqa: E402
from pg_embedding_amd.server import ServerProcess, client_lib  # noqa: E402

f32p = C.POINTER(C.c_float)
for dim in (128, 768, 1536):
    a = np.random.default_rng(1).standard_normal(dim).astype(np.float32)
    b = np.random.default_rng(2).standard_normal(dim).astype(np.float32)
    L = shim_lib()
    L.hnsw_init_dist_func()
    pa, pb = a.ctypes.data_as(f32p), b.ctypes.data_as(f32p)
    for _ in range(200):
        L.hnsw_dist_func(0, pa, pb, dim)
    t = time.perf_counter()
    n = 3000
    for _ in range(n):
        d = L.hnsw_dist_func(0, pa, pb, dim)
    us = (time.perf_counter() - t) / n * 1e6
    want = float(np.sqrt(((a.astype(np.float64) - b) ** 2).sum()))
    print(f"dim {dim}: hnsw_dist_func in process {us:.1f} us per call (value {d:.5f}, numpy {want:.5f})", flush=True)

with ServerProcess() as s:
    Lc = client_lib()
    Lc.hnsw_gpu_remote_connect(s.socket_path.encode())
    Lc.hnsw_dist_func.restype = C.c_float
    Lc.hnsw_dist_func.argtypes = [C.c_int, f32p, f32p, C.c_size_t]
    for dim in (128, 768, 1536):
        a = np.random.default_rng(1).standard_normal(dim).astype(np.float32)
        b = np.random.default_rng(2).standard_normal(dim).astype(np.float32)
        pa, pb = a.ctypes.data_as(f32p), b.ctypes.data_as(f32p)
        for _ in range(200):
            Lc.hnsw_dist_func(0, pa, pb, dim)
        t = time.perf_counter()
        n = 3000
        for _ in range(n):
            d = Lc.hnsw_dist_func(0, pa, pb, dim)
        print(f"dim {dim}: hnsw_dist_func through the server {(time.perf_counter() - t) / n * 1e6:.1f} us per call (value {d:.5f})", flush=True)
