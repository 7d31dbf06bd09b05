"""One BASELINE config on the device, for timing and for rocprofv3 passes (scripts/profile_cmd.sh):
build 1M rows, then `reps` launches of `nq` queries; prints q/s and algorithmic GB/s per launch size.
usage: exp_cfg.py <dim> <m> <metric l2|cosine> <sift 0|1> [nq,nq,...] [reps] [blocks_per_cu,...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

dim, m, metric, sift = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4])
nqs = [int(x) for x in (sys.argv[5] if len(sys.argv) > 5 else "40000").split(",")]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
bpcs = [int(x) for x in sys.argv[7].split(",")] if len(sys.argv) > 7 else [0]
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
for bpc in bpcs:
    if bpc:
        os.environ["HNSW_GPU_BLOCKS_PER_CU"] = str(bpc)
    else:
        os.environ.pop("HNSW_GPU_BLOCKS_PER_CU", None)
    for nq in nqs:
        Q = Qall[:nq].contiguous()
        out = ix.search_torch(Q, ef, stats=True)
        torch.cuda.synchronize()
        st = out["stats"].cpu().numpy().astype(np.int64)
        cnt = out["counts"].cpu().numpy().astype(np.int64)
        byt = (st[:, 0] * dim * 4 + st[:, 1] * (2 * m + 1) * 4 + dim * 4 + cnt * 8).sum()
        ms = []
        for _ in range(reps):
            ix.search_torch(Q, ef, out=out)
            ms.append(ix.last_search_ms())
        best = min(ms)
        spread = f" [min/median/max {min(ms):.3f}/{sorted(ms)[len(ms) // 2]:.3f}/{max(ms):.3f} ms over {len(ms)}]"
        import zlib
        sig = zlib.crc32(out["labels"].cpu().numpy().tobytes()) ^ zlib.crc32(out["dists"].cpu().numpy().tobytes()) ^ zlib.crc32(st.tobytes())
        stamps = ""
        if os.environ.get("HNSW_GPU_TEAM_COUNTERS"):
            c = ix.debug_counters()
            if c[0]:
                names = ("pop", "link", "visited", "score", "accept")
                stamps = " | cycles/hop " + " ".join(f"{nm} {c[1 + i] * 64 / c[0]:.0f}" for i, nm in enumerate(names)) + \
                         f" | per query: walk {c[6] * 64 / nq:.0f} emit+cleanup {c[7] * 64 / nq:.0f} hops {c[0] / nq:.1f}"
                if len(c) >= 16 and c[8]:
                    stamps += (f" | per hop: new rows {c[8] / c[0]:.2f}, below the stale bound {c[9] / c[0]:.2f}, accept-loop iterations {c[10] / c[0]:.2f}, accepted {c[11] / c[0]:.2f} "
                               f"(in the loop {c[15] / c[0]:.2f}), hops appended whole {c[12] / c[0]:.3f}, prunes {c[13] / c[0]:.4f}, hops with a second scoring pass {c[14] / c[0]:.3f}")
        print(f"dim {dim} m {m} {metric} sift={sift} nq={nq:6d} blocks/CU={bpc or 'max'} slots={ix.last_search_slots():5d} "
              f"E_q {st[:, 0].mean():.0f} H_q {st[:, 1].mean():.0f} kernel {best:8.3f} ms {nq / best * 1e3:10.0f} q/s "
              f"{byt / best / 1e6:7.0f} GB/s alg = {byt / best / 1e6 / 8000:.3f} of 8 TB/s  [{ix.last_search_kernel()}] crc {sig:08x}{stamps}{spread}", flush=True)
