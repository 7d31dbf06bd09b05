"""Experiment: BASELINE config 5 — 1M x 1536 cosine, Q=1024 batched, scoring as an f32 MFMA GEMM."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd._lib import gpu_lib
from pg_embedding_amd.datasets import gmm_torch
n, dim, nq, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
func = pg.DIST_COSINE if (len(sys.argv) < 6 or sys.argv[5] == "cosine") else pg.DIST_L2
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev); Q = gmm_torch(nq, dim, stream=1, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, 16, 64, 128, func), n); ix.append_torch(X); torch.cuda.synchronize(); del X
for rep in range(3):
    t = time.time(); i1, d1 = ix.bruteforce_torch(Q, k, mfma=True); torch.cuda.synchronize(); t1 = time.time() - t
    gemm_ms = gpu_lib().hnsw_gpu_last_bruteforce_gemm_ms()
    flops = 2.0 * nq * n * ((dim + 3) // 4 * 4)
    mhz = gpu_lib().hnsw_gpu_last_bruteforce_clock_mhz()
    tile = gpu_lib().hnsw_gpu_last_bruteforce_tile()
    print(f"mfma path ({tile} x {tile} tiles): total {t1*1e3:.1f} ms, GEMM/filter kernel {gemm_ms:.2f} ms = {flops/gemm_ms/1e9:.1f} TFLOP/s "
          f"({flops/gemm_ms/1e9/157.3:.3f} of the 157.3 TF f32 MFMA peak; shader clock in the kernel {mhz:.0f} MHz -> "
          f"{flops/gemm_ms/1e9/(157.3*mhz/2400.0):.3f} of the roof at that clock); {nq/t1:,.0f} exhaustive queries/s", flush=True)
t = time.time(); i0, d0 = ix.bruteforce_torch(Q, k); torch.cuda.synchronize(); t0 = time.time() - t
print(f"canonical scan: {t0*1e3:.1f} ms; identical ids: {bool((i0 == i1).all())}, identical distance bits: {bool((d0.view(torch.int32) == d1.view(torch.int32)).all())}")
