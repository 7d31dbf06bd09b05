"""Experiment: the other BASELINE configs as parity-style cases at scale (not bench lines):
build on device, recall@10 vs exhaustive, QPS and algorithmic GB/s, spot-check vs the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import oracle, pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch, recall_at_k

dev = torch.device("cuda", 0)
CFG = {
  "C2_sift_like_1Mx128_l2_m16":   dict(n=1_000_000, dim=128,  m=16, efc=200, ef=128, func=pg.DIST_L2,     sift=True),
  "C3_1Mx768_cosine_m32":         dict(n=1_000_000, dim=768,  m=32, efc=200, ef=128, func=pg.DIST_COSINE, sift=False),
  "C5_1Mx1536_cosine_m32":        dict(n=1_000_000, dim=1536, m=32, efc=200, ef=128, func=pg.DIST_COSINE, sift=False),
  "M_1Mx768_l2_m32":              dict(n=1_000_000, dim=768,  m=32, efc=200, ef=128, func=pg.DIST_L2,     sift=False),
  "reference_defaults_1Mx768_m100_efc16_ef64": dict(n=1_000_000, dim=768, m=100, efc=16, ef=64, func=pg.DIST_L2, sift=False),
  "manhattan_1Mx256_m16":         dict(n=1_000_000, dim=256,  m=16, efc=100, ef=128, func=pg.DIST_MANHATTAN, sift=False),
}
which = sys.argv[1:] or list(CFG)
for name in which:
    c = CFG[name]
    n, dim, m, efc, ef, func = c["n"], c["dim"], c["m"], c["efc"], c["ef"], c["func"]
    X = gmm_torch(n, dim, k=1000, sigma=0.3, seed=42, device=dev)
    Q = gmm_torch(40000, dim, k=1000, sigma=0.3, seed=42, stream=1, device=dev)
    if c["sift"]:
        X = torch.clamp(torch.round(40.0 + 35.0 * X), 0, 218); Q = torch.clamp(torch.round(40.0 + 35.0 * Q), 0, 218)
    meta = pg.make_meta(dim, m, efc, ef, func)
    ix = pg.GpuIndex.empty(meta, n); ix.append_torch(X); torch.cuda.synchronize()
    t = time.time(); ix.link(0, n); torch.cuda.synchronize(); tb = time.time() - t
    truth, _ = ix.bruteforce_torch(Q[:1000].contiguous(), 10)
    out = ix.search_torch(Q, ef, stats=True); torch.cuda.synchronize()
    rec = recall_at_k(out["labels"][:1000].cpu().numpy(), truth.cpu().numpy(), 10)
    st = out["stats"].cpu().numpy().astype(np.int64); cnt = out["counts"].cpu().numpy().astype(np.int64)
    byt = (st[:,0]*dim*4 + st[:,1]*(2*m+1)*4 + dim*4 + cnt*8)
    res = []
    for nq in (10000, 40000):
        Qs = Q[:nq].contiguous(); o = ix.search_torch(Qs, ef)
        ms = min((ix.search_torch(Qs, ef, out=o), ix.last_search_ms())[1] for _ in range(3))
        res.append(f"nq={nq}: {ms:.2f} ms {nq/ms*1e3:,.0f} QPS {byt[:nq].sum()/ms/1e6:,.0f} GB/s")
    # spot check against the CPU oracle on the same bytes (64 queries)
    port = oracle.PortIndex(dim, m, efc, ef, func, capacity=n); port.load_raw(ix.export_flat(), n)
    w = port.search_many(Q[:64].cpu().numpy(), ef, nthreads=16)
    ok = (out["labels"][:64].cpu().numpy().view(np.uint64) == w["labels"]).all() and (out["dists"][:64].cpu().numpy().view(np.uint32) == w["dists"].view(np.uint32)).all()
    print(f"{name}: build {tb:.1f}s recall@10 {rec:.4f} E_q {st[:,0].mean():.0f} H_q {st[:,1].mean():.0f} B_q {byt.mean()/1e6:.2f} MB slots {ix.last_search_slots()} | " + " | ".join(res) + f" | bit-exact vs oracle: {ok}", flush=True)
    ix.close(); del X, Q, port
