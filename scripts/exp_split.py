"""The tail split of a batch (include/hnsw_gpu.h, HNSW_GPU_SPLIT): one launch vs main launch + a tail of T queries on the library's
internal stream.  Prints the call's device time (HIP events from the main launch's start to the end of both parts), min / median
of `reps` calls, and a CRC of labels + distance bits + E_q/H_q so that every line can be seen to answer identically.
usage: exp_split.py <dim> <m> <metric l2|cosine> <sift 0|1> <nq,nq,...> <T,T,...|auto> [reps]"""
import os
import sys
import zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric, sift = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
nqs = [int(x) for x in sys.argv[5].split(",")]
tails = sys.argv[6].split(",")
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 10
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
    base_ms = None
    for T in ["0"] + tails:
        if T == "auto":
            pg.config_set("HNSW_GPU_SPLIT", None)
        else:
            pg.config_set("HNSW_GPU_SPLIT", T)
        out = ix.search_torch(Q, ef, stats=True)
        ix.search_torch(Q, ef, out=out)
        torch.cuda.synchronize()
        st = out["stats"].cpu().numpy().astype(np.int64)
        cnt = out["counts"].cpu().numpy().astype(np.int64)
        byt = (st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum()
        ms = []
        for _ in range(reps):
            ix.search_torch(Q, ef, out=out)
            ms.append(ix.last_search_ms())
        tail, tk = ix.last_search_tail()
        sig = zlib.crc32(out["labels"].cpu().numpy().tobytes()) ^ zlib.crc32(out["dists"].cpu().numpy().tobytes()) ^ zlib.crc32(st.tobytes())
        best, med = min(ms), sorted(ms)[len(ms) // 2]
        if T == "0":
            base_ms = med
        print(f"dim {dim} nq={nq:6d} split {T:>5s} -> tail {tail:5d}: call {best:7.3f} / {med:7.3f} ms (min / median of {reps}) = {nq / med * 1e3:9.0f} q/s "
              f"{byt / med / 1e6 / 8000:.3f} of 8 TB/s  ({base_ms / med:.3f}x one launch) slots {ix.last_search_slots()} "
              f"[{ix.last_search_kernel().replace('pgemb::', '')} | {tk.replace('pgemb::', '')}] crc {sig:08x}", flush=True)
