"""Where a launch's time goes between its queries: per-query start / end device-clock stamps of ONE traced launch (hnsw_gpu_search_traced_dev)
-> how many walks are in flight over the launch (ramp, plateau, drain tail), what the resident slots' utilisation is, and what the launch
would take if every slot were busy to the end.  usage: exp_timeline.py <dim> <m> <metric l2|cosine> <sift 0|1> [nq,nq,...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric, sift = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
nqs = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "40000").split(",")]
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, int(os.environ.get("EXP_EF", "128"))
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)


def rows(cnt, stream):
    X = gmm_torch(cnt, dim, stream=stream, device=dev)
    return torch.clamp(torch.round(40.0 + 35.0 * X), 0, 218) if sift else X


X = rows(n, 0)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X)
ix.link(0, n)
torch.cuda.synchronize()
del X
Qall = rows(max(nqs), 1)
for nq in nqs:
    Q = Qall[:nq].contiguous()
    out = ix.search_torch(Q, ef, stats=True)
    for _ in range(3):
        ix.search_torch(Q, ef, out=out)
    plain_ms = ix.last_search_ms()
    slots_plain = ix.last_search_slots()
    kname = ix.last_search_kernel()
    tr = ix.search_traced_torch(Q, ef, evals_cap=64)
    torch.cuda.synchronize()
    traced_ms = ix.last_search_ms()
    slots = ix.last_search_slots()
    t = tr["times"].cpu().numpy().astype(np.float64) / 100.0          # microseconds (100 MHz constant-rate clock)
    t0, t1 = t[:, 0], t[:, 1]
    base = t0.min()
    t0 -= base; t1 -= base
    span = t1.max()
    dur = t1 - t0
    work = dur.sum()
    ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    active = np.cumsum(ev[:, 1])
    # first time the number of walks in flight falls below 90 % / 50 % of the slots for good
    def last_time_at_least(frac):
        ok = np.nonzero(active >= frac * slots)[0]
        return ev[ok[-1], 0] if len(ok) else 0.0
    t90, t50 = last_time_at_least(0.9), last_time_at_least(0.5)
    print(f"dim {dim} m {m} {metric} nq={nq} kernel {kname} (timed launch: {plain_ms:.3f} ms on {slots_plain} slots; traced launch [{ix.last_search_kernel()}]: "
          f"{traced_ms:.3f} ms on {slots} slots)")
    print(f"  walk stamps: span {span / 1e3:.3f} ms; mean walk {dur.mean():.1f} us (p10 {np.percentile(dur, 10):.1f}, p50 {np.percentile(dur, 50):.1f}, "
          f"p90 {np.percentile(dur, 90):.1f}, max {dur.max():.1f}); walks per slot {nq / slots:.2f}")
    print(f"  slot utilisation by walks = {work / (slots * span):.3f}; with every slot busy to the end the walks alone would take {work / slots / 1e3:.3f} ms "
          f"({work / slots / span:.3f} of the span)")
    print(f"  >= 90 % of the slots walking until {t90 / 1e3:.3f} ms ({t90 / span:.3f} of the span), >= 50 % until {t50 / 1e3:.3f} ms ({t50 / span:.3f}); "
          f"first walk starts at 0, last start {t0.max() / 1e3:.3f} ms")
    # duration of a walk by when it started: do walks get faster as the chip empties?
    order = np.argsort(t0)
    q = len(order) // 5
    print("  mean walk (us) by start-time quintile: " + " ".join(f"{dur[order[i * q:(i + 1) * q]].mean():.1f}" for i in range(5)))
