"""A/B of one library build against another on the shapes the round's targets are stated on (run once per build, PGEMB_GPU_LIB selects it):
one query per launch, 16 / 256 / 1 024 queries, the headline launch; kernel time median / min over repeated launches, E_q, H_q and a CRC of
labels + distance bits + stats (equal between builds that walk the same walks).  usage: exp_ab.py <dim> <m> [metric] [nqs] [sift]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m = int(sys.argv[1]), int(sys.argv[2])
metric = sys.argv[3] if len(sys.argv) > 3 else "l2"
nqs = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "1,16,256,1024,10000,40000").split(",")]
sift = int(sys.argv[5]) if len(sys.argv) > 5 else 0
n, efc, ef = int(os.environ.get("EXP_ROWS", "1000000")), 200, int(os.environ.get("EXP_EF", "128"))
func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
dev = torch.device("cuda", 0)


def rows(cnt, stream):
    X = gmm_torch(cnt, dim, stream=stream, device=dev)
    return torch.clamp(torch.round(40.0 + 35.0 * X), 0, 218) if sift else X


X = rows(n, 0)
ix = pg.GpuIndex.empty(pg.make_meta(dim, m, efc, ef, func), n)
ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize(); del X
Qall = rows(max(max(nqs), 64), 1)
for nq in nqs:
    ms, crc = [], 0
    if nq == 1:
        for i in range(56):
            out = ix.search_torch(Qall[i:i + 1].contiguous(), ef, stats=True)
            torch.cuda.synchronize()
            ms.append(ix.last_search_ms())
            st = out["stats"].cpu().numpy()
            crc ^= zlib.crc32(out["labels"].cpu().numpy().tobytes()) ^ zlib.crc32(out["dists"].cpu().numpy().tobytes()) ^ zlib.crc32(st.tobytes())
        ms = ms[8:]
    else:
        Q = Qall[:nq].contiguous()
        out = ix.search_torch(Q, ef, stats=True)
        for _ in range(8):
            ix.search_torch(Q, ef, out=out); torch.cuda.synchronize()
            ms.append(ix.last_search_ms())
        st = out["stats"].cpu().numpy()
        crc = zlib.crc32(out["labels"].cpu().numpy().tobytes()) ^ zlib.crc32(out["dists"].cpu().numpy().tobytes()) ^ zlib.crc32(st.tobytes())
    byt = (st[:, 0].astype(np.int64) * dim * 4 + st[:, 1].astype(np.int64) * (2 * m + 1) * 4 + dim * 4 + ef * 8).sum()
    print(f"dim {dim} m {m} {metric} nq {nq:6d}: kernel median {np.median(ms):8.4f} ms min {min(ms):8.4f} ms  {nq / np.median(ms) * 1e3:10.0f} q/s  "
          f"{byt / np.median(ms) / 1e6 / 8000:.3f} of 8 TB/s  E_q {st[:, 0].mean():7.1f} H_q {st[:, 1].mean():6.1f}  crc {crc:08x}  [{ix.last_search_kernel()}]", flush=True)
