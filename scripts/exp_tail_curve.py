"""A launch of ONE walk per walking wave (Q = 1 024, team form): where does its time go?  Every walk starts at 0; a walk of h hops that ends at
t says the launch had advanced h hops by t, so (t_end, hops) over the walks of ONE traced launch is the progress curve H(t) of the launch, its
slope the hop time by phase: all walks in flight (the memory system loaded) -> few walks left, each with up to seven helpers.
Prints the curve in ten time slices, the walks in flight at each, and the launch time under HNSW_GPU_TEAM_SPEC = default / 0 / 2 / 8.
usage: exp_tail_curve.py <dim> <m> <metric l2|cosine> [nq=1024]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
nq = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, 128
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, stream=0, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Q = gmm_torch(nq, dim, stream=1, device=dev)
out = ix.search_torch(Q, ef, stats=True)
torch.cuda.synchronize()
st = out["stats"].cpu().numpy().astype(np.int64)
cnt = out["counts"].cpu().numpy().astype(np.int64)
byt = float((st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum())
hops, evals = st[:, 1], st[:, 0]
print(f"dim {dim} m {m} {metric} nq {nq}: hops mean {hops.mean():.1f} p10 {np.percentile(hops, 10):.0f} p50 {np.percentile(hops, 50):.0f} p90 {np.percentile(hops, 90):.0f} "
      f"p99 {np.percentile(hops, 99):.0f} max {hops.max()}; evals mean {evals.mean():.0f} max {evals.max()}; {byt / 1e9:.3f} GB algorithmic per launch", flush=True)


def timed(reps=7):
    ms = []
    for _ in range(reps):
        ix.search_torch(Q, ef, out=out)
        ms.append(ix.last_search_ms())
    return float(np.median(ms[1:])), float(np.min(ms))


for spec in (None, 0, 2, 8):
    pg.config_set("HNSW_GPU_TEAM_SPEC", spec)
    med, best = timed()
    print(f"  HNSW_GPU_TEAM_SPEC={spec}: launch median {med:.3f} ms (min {best:.3f}) = {byt / (med * 1e-3) / 8e12:.3f} of 8 TB/s  [{ix.last_search_kernel()}, {ix.last_search_slots()} slots]", flush=True)
pg.config_set("HNSW_GPU_TEAM_SPEC", None)

tr = ix.search_traced_torch(Q, ef, evals_cap=64)
torch.cuda.synchronize()
t = tr["times"].cpu().numpy().astype(np.float64) / 100.0            # us
t0, t1 = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
span = t1.max()
print(f"traced launch {ix.last_search_ms():.3f} ms, span of the walk stamps {span / 1e3:.3f} ms, last start {t0.max():.1f} us; mean walk {np.mean(t1 - t0):.1f} us")
order = np.argsort(t1)
te, he = t1[order], hops[order].astype(np.float64)
# progress curve: a running median of the hops of the walks that end around t
edges = np.linspace(te[0], span, 11)
prev_t, prev_h = 0.0, 0.0
print("  slice end (us) | walks ending in it | still walking at its end | hops of the walks ending in it (median) | us per hop over the slice")
for i in range(10):
    sel = (te > edges[i]) & (te <= edges[i + 1]) if i else (te <= edges[1])
    if not sel.any():
        continue
    hmed = float(np.median(he[sel]))
    tmid = float(np.median(te[sel]))
    rate = (tmid - prev_t) / max(hmed - prev_h, 1e-9)
    print(f"  {edges[i + 1]:10.1f} | {int(sel.sum()):6d} | {int((te > edges[i + 1]).sum()):6d} | {hmed:7.1f} | {rate:6.2f}")
    prev_t, prev_h = tmid, hmed
# the longest walks
print("  the 10 walks that end last: end us | hops | evals | us per hop of the whole walk")
for i in order[-10:]:
    print(f"   {t1[i]:8.1f} | {hops[i]:4d} | {evals[i]:5d} | {t1[i] / hops[i]:.2f}")
r = np.corrcoef(t1, hops)[0, 1]
print(f"  corr(end time, hops) = {r:.3f}; fit end = a + b * hops: b = {np.polyfit(hops, t1, 1)[0]:.2f} us per hop, a = {np.polyfit(hops, t1, 1)[1]:.1f} us")
