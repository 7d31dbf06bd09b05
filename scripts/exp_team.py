"""Team form vs one-wave form: kernel time per launch size, outputs compared bit for bit.
usage: exp_team.py <dim> <m> [metric] [sift]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m = int(sys.argv[1]), int(sys.argv[2])
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
sift = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, 128
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
Qall = rows(40000, 1)
variants = [("one-wave", {"HNSW_GPU_TEAM": "0"}),
            ("team wpb8", {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "8"}),
            ("team wpb4", {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "4"}),
            ("team wpb2", {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "2"})]
ref = {}
for nq in [int(x) for x in os.environ.get("EXP_NQS", "1,16,256,1024,2560,10000,40000").split(",")]:
    for name, env in variants:
        for k in ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_WPB"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ms = []
        if nq == 1:
            for i in range(40):
                out = ix.search_torch(Qall[i:i + 1].contiguous(), ef, stats=True)
                ms.append(ix.last_search_ms())
            out = ix.search_torch(Qall[:1].contiguous(), ef, stats=True)
            t = float(np.median(ms[4:]))
        else:
            Q = Qall[:nq].contiguous()
            out = ix.search_torch(Q, ef, stats=True)
            for _ in range(4):
                ix.search_torch(Q, ef, out=out)
                ms.append(ix.last_search_ms())
            t = min(ms)
        torch.cuda.synchronize()
        cur = (out["labels"].cpu().numpy(), out["dists"].cpu().numpy().view(np.uint32), out["stats"].cpu().numpy())
        if name == "one-wave":
            ref[nq] = cur
            same = True
        else:
            same = all((x == y).all() for x, y in zip(ref[nq], cur))
        cnt = ""
        if os.environ.get("HNSW_GPU_TEAM_COUNTERS") and name != "one-wave":
            c = ix.team_counters()
            cnt = (f" | hops {c['hops']} with helpers {c['hops_with_helpers']} link hits {c['link_hits']} ids {c['ids_looked_up']} "
                   f"dist hits {c['dist_hits']} hops that scored {c['hops_that_scored']} waited {c['hops_that_waited']} polls {c['wait_polls']} "
                   f"cyc pop+links {c['cyc_pop_links']} dists {c['cyc_dists']} accept {c['cyc_accept']} | helpers: {c['helper_elements']} elements, "
                   f"{c['helper_cycles'] // max(c['helper_elements'], 1)} cyc each")
        print(f"dim {dim} nq={nq:6d} {name:10s} kernel {t:8.3f} ms {nq / t * 1e3:10.0f} q/s slots {ix.last_search_slots():5d} "
              f"identical={same} [{ix.last_search_kernel()}]{cnt}", flush=True)
