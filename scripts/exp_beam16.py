"""Experiment: ef in (256, 512] — beam form with 16 set registers vs the generic LDS form (same index)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch
dev = torch.device("cuda", 0); m, efc = 16, 200
for dim in (768, 128):
    n = 1_000_000
    X = gmm_torch(n, dim, device=dev)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, 128, pg.DIST_L2), n); ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize(); del X
    Q = gmm_torch(20000, dim, stream=1, device=dev)
    for ef in (300, 400, 512):
        ref = None
        for b in ("0", "1"):
            os.environ["HNSW_GPU_BEAM16"] = b
            out = ix.search_torch(Q, ef, stats=True); torch.cuda.synchronize()
            cur = [out[k].cpu().numpy() for k in ("labels", "counts", "stats")] + [out["dists"].cpu().numpy().view(np.uint32)]
            same = "" if ref is None else f" identical: {all((x == y).all() for x, y in zip(ref, cur))}"
            ref = ref or cur
            ms = min((ix.search_torch(Q, ef, out=out), ix.last_search_ms())[1] for _ in range(3))
            print(f"dim {dim} ef {ef} {'beam16' if b == '1' else 'LDS   '}: {ms:8.2f} ms {20000/ms*1e3:10,.0f} q/s slots {ix.last_search_slots()}{same}", flush=True)
    ix.close()
