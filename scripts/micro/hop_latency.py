"""Per-hop latency of ONE query on an idle GPU vs index size: a tiny index lives in L2, so the difference to a
1M-row index is the part of the hop that waits for HBM; the rest is the dependent instruction chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch
dev = torch.device("cuda", 0)
for dim in (128, 768):
    for n in (2000, 50000, 1000000):
        X = gmm_torch(n, dim, k=min(1000, n // 20), device=dev)
        ix = pg.GpuIndex.empty(pg.make_meta(dim, 16, 200, 128, pg.DIST_L2), n); ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize()
        Q = gmm_torch(64, dim, k=min(1000, n // 20), stream=1, device=dev)
        hops, evals, ms = [], [], []
        for i in range(64):
            q = Q[i:i + 1].contiguous()
            out = ix.search_torch(q, 128, stats=True); torch.cuda.synchronize()
            t = min((ix.search_torch(q, 128, out=out), ix.last_search_ms())[1] for _ in range(3))
            st = out["stats"].cpu().numpy()
            evals.append(int(st[0, 0])); hops.append(int(st[0, 1])); ms.append(t)
        h, e, t = np.mean(hops), np.mean(evals), np.mean(ms)
        print(f"dim {dim} n {n:8d}: {t*1e3:7.1f} us/query, {h:6.1f} hops, {e:7.1f} evals -> {t*1e3/h:5.2f} us/hop", flush=True)
        ix.close()
