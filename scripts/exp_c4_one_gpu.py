"""BASELINE config C4 functionally at full size on ONE device: 10M x 768 L2 as 8 row shards
(1.25M rows each, independent graphs), per-shard searchKnn + merge kernel, recall vs exhaustive.
(On an 8-GPU node the shards live on different GPUs and the lists travel by one RCCL all-gather:
bench.py --mode sharded.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch, recall_at_k
from pg_embedding_amd.sharded import shard_range

n, dim, shards, nq, ef = int(sys.argv[1]), 768, 8, 1024, 128
dev = torch.device("cuda", 0)
meta = pg.make_meta(dim, 16, 200, ef, pg.DIST_L2)
Q = gmm_torch(nq, dim, stream=1, device=dev)
idx, t_build = [], 0.0
for r in range(shards):
    lo, hi = shard_range(n, shards, r)
    rows = gmm_torch(hi - lo, dim, stream=100 + r, device=dev)
    ix = pg.GpuIndex.empty(meta, hi - lo)
    ix.append_torch(rows, torch.arange(lo, hi, dtype=torch.int64, device=dev))
    torch.cuda.synchronize(); t = time.time(); ix.link(0, hi - lo); torch.cuda.synchronize(); t_build += time.time() - t
    idx.append(ix); del rows
def run():
    outs = [ix.search_torch(Q, ef) for ix in idx]
    L = torch.stack([o["labels"] for o in outs]).contiguous(); D = torch.stack([o["dists"] for o in outs]).contiguous()
    return pg.merge_topk_torch(L, D, ef)
ml, md, mc = run(); torch.cuda.synchronize()
t = time.time()
for _ in range(3): ml, md, mc = run()
torch.cuda.synchronize(); dt = (time.time() - t) / 3
# exhaustive truth per shard (MFMA), merged on the host
cand_i, cand_d = [], []
for r, ix in enumerate(idx):
    lo, _ = shard_range(n, shards, r)
    ti, td = ix.bruteforce_torch(Q, 10, mfma=True)
    cand_i.append(ti.long() + lo); cand_d.append(td)
ci = torch.cat(cand_i, 1); cd = torch.cat(cand_d, 1)
order = torch.argsort(cd, dim=1)[:, :10]
truth = torch.gather(ci, 1, order)
rec = recall_at_k(ml.cpu().numpy(), truth.cpu().numpy(), 10)
print(f"C4 on one device: {n}x{dim} in {shards} shards, build {t_build:.1f}s total; Q={nq} ef={ef}: "
      f"{dt*1e3:.1f} ms per batch over all shards + merge = {nq/dt:,.0f} q/s (one GPU doing the work of 8); "
      f"recall@10 {rec:.4f}; counts full: {bool((mc == ef).all())}")
