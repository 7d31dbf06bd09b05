"""Is the slow state of the narrow-row launch (8.25 ms instead of 5.85 ms per 40 000 queries: bench run r5ac, round 3's r3ag) the device's power management
after sustained load?  Build both indexes, time the narrow-row launch fresh, then right after <seconds> of back-to-back wide-row launches, then after idle.
usage: exp_throttle.py [seconds of load, default 15]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

load_s = float(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 15.0
dev = torch.device("cuda", 0)
ef = 128


def build(dim, rows):
    ix = pg.GpuIndex.empty(pg.make_meta(dim, 16, 200, ef, pg.DIST_L2), rows.shape[0])
    ix.append_torch(rows); ix.link(0, rows.shape[0]); torch.cuda.synchronize()
    return ix


def timed(ix, Q, out, reps):
    ms = []
    for _ in range(reps):
        ix.search_torch(Q, ef, out=out); torch.cuda.synchronize(); ms.append(ix.last_search_ms())
    return "median %.3f ms (min %.3f max %.3f)" % (float(np.median(ms)), min(ms), max(ms))


Xn = torch.clamp(torch.round(40.0 + 35.0 * gmm_torch(1_000_000, 128, device=dev)), 0, 218)
ixn = build(128, Xn); del Xn
Qn = torch.clamp(torch.round(40.0 + 35.0 * gmm_torch(40000, 128, stream=1, device=dev)), 0, 218)
Xw = gmm_torch(1_000_000, 768, device=dev)
ixw = build(768, Xw); del Xw
Qw = gmm_torch(40000, 768, stream=1, device=dev)
on = ixn.search_torch(Qn, ef, stats=True); ow = ixw.search_torch(Qw, ef, stats=True)
time.sleep(3.0)
print("narrow rows, after 3 s of idle      :", timed(ixn, Qn, on, 8), flush=True)
print("wide rows,   after the narrow ones   :", timed(ixw, Qw, ow, 4), flush=True)
t0 = time.time(); k = 0
while time.time() - t0 < load_s:
    ixw.search_torch(Qw, ef, out=ow); k += 1
torch.cuda.synchronize()
print(f"({k} wide-row launches back to back in {time.time() - t0:.1f} s; the last: {ixw.last_search_ms():.3f} ms)", flush=True)
print("narrow rows, right after that load   :", timed(ixn, Qn, on, 12), flush=True)
print("narrow rows, 12 more launches        :", timed(ixn, Qn, on, 12), flush=True)
time.sleep(5.0)
print("narrow rows, after 5 s of idle       :", timed(ixn, Qn, on, 8), flush=True)
print("wide rows,   after that              :", timed(ixw, Qw, ow, 4), flush=True)
