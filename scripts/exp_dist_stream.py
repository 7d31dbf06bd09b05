"""Roofline of the batched distance kernel alone (hnsw_dist_func over many rows): 1M x dim rows streamed once."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import torch
import pg_embedding_amd as pg
from pg_embedding_amd._lib import gpu_lib, check
L = gpu_lib(); dev = torch.device("cuda", 0)
for dim in (128, 768, 1536):
    n = 2_000_000 if dim <= 768 else 1_000_000
    X = torch.randn((n, dim), device=dev); q = torch.randn(dim, device=dev); out = torch.empty(n, device=dev)
    for func, name in ((0, "l2"), (1, "cosine"), (2, "manhattan")):
        s = torch.cuda.current_stream().cuda_stream
        check(L.hnsw_gpu_dist_batch_dev(func, q.data_ptr(), X.data_ptr(), n, dim, dim, out.data_ptr(), s), "dist")
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            check(L.hnsw_gpu_dist_batch_dev(func, q.data_ptr(), X.data_ptr(), n, dim, dim, out.data_ptr(), s), "dist")
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"dist_batch {name:9s} {n}x{dim}: {ms:.3f} ms  {n*dim*4/ms/1e6:,.0f} GB/s ({n*dim*4/ms/1e6/8000:.2f} of 8 TB/s)", flush=True)
    del X
