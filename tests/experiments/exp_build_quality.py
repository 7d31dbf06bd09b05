"""Experiment: batched device build vs serial (reference-order) build — degree, E_q, recall."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import oracle, pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch, recall_at_k

n, dim, m, efc, ef = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), 128
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev); Q = gmm_torch(2000, dim, stream=1, device=dev)
meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)

def report(tag, ix, secs):
    raw = ix.export_flat().reshape(n, -1)
    cnt = raw[:, :4].copy().view(np.uint32).ravel()
    truth, _ = ix.bruteforce_torch(Q[:1000].contiguous(), 10)
    out = ix.search_torch(Q, ef, stats=True); torch.cuda.synchronize()
    st = out["stats"].cpu().numpy()
    rec = recall_at_k(out["labels"][:1000].cpu().numpy(), truth.cpu().numpy(), 10)
    print(f"{tag}: build {secs:.1f}s deg mean {cnt.mean():.1f} max {cnt.max()} E_q {st[:,0].mean():.0f} H_q {st[:,1].mean():.0f} recall {rec:.4f}", flush=True)

for mb, ratio in [(0, 0), (4096, 32), (512, 64)]:
    ix = pg.GpuIndex.empty(meta, n); ix.append_torch(X); torch.cuda.synchronize()
    t = time.time(); ix.link(0, n, mb, ratio); torch.cuda.synchronize()
    report(f"gpu batched max_batch={mb} ratio={ratio}", ix, time.time() - t); ix.close()
if len(sys.argv) > 5:
    t = time.time(); cpu = oracle.RefIndex(dim, m, efc, ef, pg.DIST_L2, capacity=n); cpu.add(X.cpu().numpy()); s = time.time() - t
    ix = pg.GpuIndex.from_flat(meta, cpu.raw(), n)
    report("cpu serial (reference code)", ix, s)
