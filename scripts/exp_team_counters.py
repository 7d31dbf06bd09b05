"""Team-form counters of a launch (needs a -DHNSW_TEAM_COUNTERS build: PGEMB_GPU_LIB=.../libhnsw_gpu_teamcnt.so, HNSW_GPU_TEAM_COUNTERS=1):
how many hops found a helper's package, how many waited for one and for how long, how many still scored rows themselves, and what a
helper's package costs.   usage: exp_team_counters.py <dim> <m> [metric] [nqs]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m = int(sys.argv[1]), int(sys.argv[2])
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
nqs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,1024").split(",")]
n, efc, ef = 1_000_000, 200, 128
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize(); del X
Qall = gmm_torch(max(max(nqs), 64), dim, stream=1, device=dev)
for nq in nqs:
    reps = 32 if nq == 1 else 4
    tot = None
    for i in range(reps):
        Q = Qall[i:i + 1].contiguous() if nq == 1 else Qall[:nq].contiguous()
        out = ix.search_torch(Q, ef, stats=True); torch.cuda.synchronize()
        c = ix.team_counters()
        tot = c if tot is None else {k: tot[k] + v for k, v in c.items()}
    h = max(tot["hops"], 1)
    print(f"dim {dim} {metric} nq {nq}: hops {h // reps} per launch; with helpers {tot['hops_with_helpers'] / h:.3f}; link list from a package {tot['link_hits'] / h:.3f}; "
          f"hops that waited for a claimed package {tot['hops_that_waited'] / h:.3f} ({tot['wait_polls'] / max(tot['hops_that_waited'], 1):.0f} polls each); "
          f"new rows {tot['ids_looked_up'] / h:.2f} per hop, distance from a package {tot['dist_hits'] / max(tot['ids_looked_up'], 1):.3f}; hops that still scored rows {tot['hops_that_scored'] / h:.3f}; "
          f"cycles per hop: pop+publish+look-up+links {tot['cyc_pop_links'] / h:.0f}, visited+distances {tot['cyc_dists'] / h:.0f}, accept {tot['cyc_accept'] / h:.0f}; "
          f"helpers: {tot['helper_elements'] / h:.2f} packages per hop, {tot['helper_cycles'] / max(tot['helper_elements'], 1):.0f} cycles each  [{ix.last_search_kernel()}]", flush=True)
