"""Characterisation: efsearch sweep at 1M x 768 and row-count sweep at efsearch=128 (L2, m=16, efc=200)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch, recall_at_k
dev = torch.device("cuda", 0); dim, m, efc = 768, 16, 200
def build(n):
    X = gmm_torch(n, dim, device=dev)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, 128, pg.DIST_L2), n); ix.append_torch(X); torch.cuda.synchronize()
    t = time.time(); ix.link(0, n); torch.cuda.synchronize()
    return ix, time.time() - t
def measure(ix, Q, ef, nq):
    Qs = Q[:nq].contiguous()
    out = ix.search_torch(Qs, ef, stats=True); torch.cuda.synchronize()
    st = out["stats"].cpu().numpy().astype(np.int64); cnt = out["counts"].cpu().numpy().astype(np.int64)
    byt = (st[:,0]*dim*4 + st[:,1]*(2*m+1)*4 + dim*4 + cnt*8).sum()
    ms = min((ix.search_torch(Qs, ef, out=out), ix.last_search_ms())[1] for _ in range(3))
    return out, st, ms, byt
Q = gmm_torch(40000, dim, stream=1, device=dev)
ix, tb = build(1_000_000)
truth, _ = ix.bruteforce_torch(Q[:1000].contiguous(), 10, mfma=True)
print(f"# efsearch sweep, 1M x 768 (build {tb:.1f}s), 40000 queries/launch")
for ef in (16, 32, 64, 128, 256, 512):
    out, st, ms, byt = measure(ix, Q, ef, 40000)
    rec = recall_at_k(out["labels"][:1000].cpu().numpy(), truth.cpu().numpy(), min(10, ef))
    print(f"ef={ef:4d}: recall@10 {rec:.4f} E_q {st[:,0].mean():.0f} H_q {st[:,1].mean():.0f} {ms:.2f} ms {40000/ms*1e3:,.0f} q/s {byt/ms/1e6:,.0f} GB/s alg ({'register' if ef <= 256 else 'LDS'} form)", flush=True)
ix.close()
print("# row-count sweep, efsearch=128, 40000 queries/launch")
for n in (100_000, 1_000_000, 4_000_000, 10_000_000):
    ix, tb = build(n)
    truth, _ = ix.bruteforce_torch(Q[:500].contiguous(), 10, mfma=True)
    out, st, ms, byt = measure(ix, Q, 128, 40000)
    rec = recall_at_k(out["labels"][:500].cpu().numpy(), truth.cpu().numpy(), 10)
    print(f"n={n:9d}: build {tb:5.1f}s ({n/tb:,.0f} inserts/s) recall@10 {rec:.4f} E_q {st[:,0].mean():.0f} H_q {st[:,1].mean():.0f} {ms:.2f} ms {40000/ms*1e3:,.0f} q/s {byt/ms/1e6:,.0f} GB/s alg", flush=True)
    ix.close()
