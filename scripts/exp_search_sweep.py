"""Experiment: search throughput vs batch size and resident blocks per CU (one build)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch

n, dim, m, efc, ef = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
dev = torch.device("cuda", 0)
X = gmm_torch(n, dim, device=dev)
meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
ix = pg.GpuIndex.empty(meta, n); ix.append_torch(X); ix.link(0, n); torch.cuda.synchronize(); del X
Qall = gmm_torch(160000, dim, stream=1, device=dev)
def run(nq, bpc, reps=3):
    if bpc: os.environ["HNSW_GPU_BLOCKS_PER_CU"] = str(bpc)
    else: os.environ.pop("HNSW_GPU_BLOCKS_PER_CU", None)
    Q = Qall[:nq].contiguous()
    out = ix.search_torch(Q, ef, stats=True); torch.cuda.synchronize()
    st = out["stats"].cpu().numpy().astype(np.int64); cnt = out["counts"].cpu().numpy().astype(np.int64)
    byt = (st[:,0]*dim*4 + st[:,1]*(2*m+1)*4 + dim*4 + cnt*8).sum()
    ms = []
    for _ in range(reps):
        ix.search_torch(Q, ef, out=out); ms.append(ix.last_search_ms())
    ms = min(ms)
    print(f"nq={nq:6d} blocks/CU={bpc or 'max'} slots={ix.last_search_slots():5d} kernel {ms:8.2f} ms  {nq/ms*1e3:10.0f} QPS  {byt/ms/1e6:7.0f} GB/s alg", flush=True)
mode = sys.argv[6] if len(sys.argv) > 6 else "full"
if mode == "full":
    for nq in (1, 256, 2560, 5120, 10000, 20000, 40000, 160000):
        run(nq, 0)
    for bpc in (1, 2, 3, 4, 5):
        run(40000, bpc)
elif mode == "shape":
    for sh in ("12x2", "12x1"):
        if sh == "12x1": os.environ["HNSW_GPU_SHAPE_12X1"] = "1"
        else: os.environ.pop("HNSW_GPU_SHAPE_12X1", None)
        print("shape", sh, flush=True)
        for nq in (1, 2048, 10000, 40000, 160000):
            run(nq, 0, reps=5)
elif mode == "beam":
    efs = [int(x) for x in (sys.argv[7].split(",") if len(sys.argv) > 7 else [str(ef)])]
    for e in efs:
        ef = e
        ref = None
        for b in ("0", "1"):
            os.environ["HNSW_GPU_BEAM"] = b
            print("ef", ef, "beam form" if b == "1" else "register form", flush=True)
            o = ix.search_torch(Qall[:20000].contiguous(), ef, stats=True); torch.cuda.synchronize()
            cur = (o["labels"].cpu().numpy(), o["dists"].cpu().numpy().view(np.uint32), o["counts"].cpu().numpy(), o["stats"].cpu().numpy())
            if ref is None: ref = cur
            else: print("   identical to register form:", all((x == y).all() for x, y in zip(ref, cur)), flush=True)
            for nq in (1, 10000, 40000, 160000):
                run(nq, 0, reps=5)
elif mode == "env":
    # A/B of one environment knob: exp_search_sweep.py n dim m efc ef env NAME v1,v2,...
    name, vals = sys.argv[7], sys.argv[8].split(",")
    ref = None
    for v in vals:
        os.environ[name] = v
        print(name, "=", v, flush=True)
        o = ix.search_torch(Qall[:20000].contiguous(), ef, stats=True); torch.cuda.synchronize()
        cur = (o["labels"].cpu().numpy(), o["dists"].cpu().numpy().view(np.uint32), o["counts"].cpu().numpy(), o["stats"].cpu().numpy())
        if ref is None: ref = cur
        else: print("   identical to the first setting:", all((x == y).all() for x, y in zip(ref, cur)), flush=True)
        for nq in (1, 10000, 40000, 160000):
            run(nq, 0, reps=5)
elif mode == "hash":
    for h in (0, 1024, 2048, 4096):
        os.environ["HNSW_GPU_HASH_ENTRIES"] = str(h)
        print("hash entries", h, flush=True)
        for nq in (1, 10000, 40000):
            run(nq, 0, reps=5)
else:
    for bpc in (1, 2, 3, 4, 5):
        for nq in (10000, 40000):
            run(nq, bpc, reps=5)
