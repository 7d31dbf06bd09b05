"""Narrow rows (config C2): what are the walks that end a launch?  One traced launch (per-query start / end stamps, E_q, H_q): a walk's
duration regressed on its hops and evaluations over the walks of the steady state, then the longest walks and the walks that END last
against that model — is the drain made of ordinary long walks (inherent), or of walks that are slow per hop (fixable)?
usage: exp_walk_tail.py [nq=40000]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
import bench

nq = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
dev = torch.device("cuda", 0)
sys.argv = sys.argv[:1]
args = bench.parse()
args.n, args.efc, args.ef, args.max_batch, args.ratio = 1_000_000, 200, 128, 0, 0
case = list(bench.side_cases(dev)[0])
case[5] = nq
ix, Q = bench.build_side_config(args, tuple(case), dev, 0)
out = ix.search_torch(Q, 128, stats=True)
for _ in range(3):
    ix.search_torch(Q, 128, out=out)
plain_ms = ix.last_search_ms()
tr = ix.search_traced_torch(Q, 128, evals_cap=64)
torch.cuda.synchronize()
traced_ms = ix.last_search_ms()
slots = ix.last_search_slots()
t = tr["times"].cpu().numpy().astype(np.float64) / 100.0          # microseconds
st = tr["stats"].cpu().numpy().astype(np.float64)
E, H = st[:, 0], st[:, 1]
t0, t1 = t[:, 0] - t[:, 0].min(), t[:, 1] - t[:, 0].min()
dur = t1 - t0
span = t1.max()
print(f"timed launch {plain_ms:.3f} ms, traced launch {traced_ms:.3f} ms on {slots} slots [{ix.last_search_kernel()}], span of the walk stamps {span / 1e3:.3f} ms")
print(f"E_q mean {E.mean():.0f} p50 {np.percentile(E, 50):.0f} p90 {np.percentile(E, 90):.0f} p99 {np.percentile(E, 99):.0f} p99.9 {np.percentile(E, 99.9):.0f} max {E.max():.0f}; "
      f"H_q mean {H.mean():.1f} p90 {np.percentile(H, 90):.0f} p99 {np.percentile(H, 99):.0f} max {H.max():.0f}")
steady = (t0 > 0.1 * span) & (t1 < 0.7 * span)              # walks that ran entirely while every slot was busy
A = np.stack([H[steady], E[steady], np.ones(steady.sum())], 1)
coef, *_ = np.linalg.lstsq(A, dur[steady], rcond=None)
pred = H * coef[0] + E * coef[1] + coef[2]
res = dur[steady] - pred[steady]
print(f"steady state ({steady.sum()} walks): duration = {coef[0]:.3f} us/hop + {coef[1]:.4f} us/evaluation + {coef[2]:.1f} us; residual std {res.std():.1f} us "
      f"(mean walk {dur[steady].mean():.1f} us); quadratic term in E: " +
      f"{np.linalg.lstsq(np.stack([H[steady], E[steady], E[steady] ** 2, np.ones(steady.sum())], 1), dur[steady], rcond=None)[0][2]:.3e} us/eval^2")
print("the 15 longest walks: duration us | hops | evaluations | model | start ms")
for q in np.argsort(-dur)[:15]:
    print(f"   {dur[q]:8.1f} | {H[q]:4.0f} | {E[q]:5.0f} | {pred[q]:8.1f} | {t0[q] / 1e3:.3f}")
print("the 15 walks that END last: end ms | start ms | duration us | hops | evaluations | model (loaded speed)")
for q in np.argsort(-t1)[:15]:
    print(f"   {t1[q] / 1e3:.3f} | {t0[q] / 1e3:.3f} | {dur[q]:8.1f} | {H[q]:4.0f} | {E[q]:5.0f} | {pred[q]:8.1f}")
# what the launch would take with the same walks but every walk starting later than t_x being as fast as the model at lone speed...
last_start = t0.max()
print(f"last walk starts at {last_start / 1e3:.3f} ms; the launch ends {(span - last_start) / 1e3:.3f} ms later; walks in flight at the last start: {((t0 <= last_start) & (t1 > last_start)).sum()}")
late = t0 > last_start - 300.0
print(f"walks started in the last 0.3 ms before the ticket ran out: {late.sum()}, their durations p50 {np.percentile(dur[late], 50):.0f} p90 {np.percentile(dur[late], 90):.0f} max {dur[late].max():.0f} us "
      f"(steady state p50 {np.percentile(dur[steady], 50):.0f} p90 {np.percentile(dur[steady], 90):.0f} max {dur[steady].max():.0f})")
