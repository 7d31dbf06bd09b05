#!/usr/bin/env python3
"""bench.py — queries/sec of the HNSW search hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch of `--nq` (default 40 000) synthetic queries
(already resident in HBM): the fused searchBaseLayer/searchKnn kernel over an HBM-resident index.
Workload at N=1 = the configuration the metric is quoted on: 1M x 768 fp32, L2,
efsearch=128 (graph built in HBM by the device insert path before the timed region).
N>1: one process per GPU, every rank holds a replica of the index and its own query
batch (queries are the independent units; no data-path collective) -> weak scaling.

Prints ONE JSON line on rank 0 (see the contract in the task description) with two extra
objects: "roofline" (dominant kernel vs the HBM roof, from HIP events on the kernel's own
stream) and "cpu_baseline" (the reference's CPU path on the same graph bytes, bounded
sample, host cores of this box).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=1_000_000, help="index rows")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--hnsw-m", dest="m", type=int, default=16, help="m reloption (maxM = 2m)")
    ap.add_argument("--efc", type=int, default=200, help="efconstruction for the device build")
    ap.add_argument("--ef", type=int, default=128, help="efsearch")
    ap.add_argument("--nq", type=int, default=40_000, help="queries per step per GPU")
    ap.add_argument("--nq-small", type=int, default=10_000, help="also report a smaller launch (0 = skip)")
    ap.add_argument("--metric", default="l2", choices=["l2", "cosine", "manhattan"])
    ap.add_argument("--clusters", type=int, default=1000)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--ratio", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded"],
                    help="replicas: index mirrored on every GPU, queries sharded (headline metric). sharded: rows partitioned "
                         "across GPUs (config C4), every GPU searches every query, RCCL all-gather + merge kernel")
    return ap.parse_args()


def main():
    args = parse()
    if args.mode == "sharded":
        return main_sharded(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ       # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- index: synthetic rows generated in HBM, graph built by the device insert path ----
    t0 = time.time()
    X = gmm_torch(args.n, args.dim, k=args.clusters, sigma=0.3, seed=42, device=dev)
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    ix = pg.GpuIndex.empty(meta, args.n, device=local)
    ix.append_torch(X)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    ix.link(0, args.n, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    del X

    # every rank searches its own query stream (weak scaling)
    Q = gmm_torch(args.nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1 + rank, device=dev)

    # ---- recall@10 against exhaustive search with the same metric ---------------------
    nrec = min(args.recall_queries, args.nq)
    truth, _ = ix.bruteforce_torch(Q[:nrec].contiguous(), 10)
    out = ix.search_torch(Q, args.ef, stats=True)
    torch.cuda.synchronize()
    labels0 = out["labels"].clone()
    recall = recall_at_k(labels0[:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
    stats = out["stats"].cpu().numpy().astype(np.int64)
    counts = out["counts"].cpu().numpy().astype(np.int64)
    E, H, R = stats[:, 0], stats[:, 1], counts
    maxM = 2 * args.m
    # algorithmic bytes per query, SURVEY.md §8(d):
    #   B_q = E_q*dim*4 + H_q*(maxM+1)*4 + dim*4 + R_q*8
    bytes_q = E * args.dim * 4 + H * (maxM + 1) * 4 + args.dim * 4 + R * 8
    bytes_launch = float(bytes_q.sum())

    # ---- the same hot path at a smaller launch (the fixed ramp-up/drain cost of a launch is
    # amortised over fewer queries); done BEFORE the timed region so that the last launches of
    # the process — the ones profiles/ summarises — are the headline workload
    small = None
    if args.nq_small and args.nq_small < args.nq:
        Qs = Q[:args.nq_small].contiguous()
        bs = ix.search_torch(Qs, args.ef)
        ms_small = []
        for _ in range(3):
            ix.search_torch(Qs, args.ef, out=bs)
            ms_small.append(ix.last_search_ms())
        small = {"queries_per_launch": args.nq_small, "kernel_ms": float(np.median(ms_small)),
                 "queries_per_s": args.nq_small / float(np.median(ms_small)) * 1e3,
                 "achieved_GBps": float(bytes_q[:args.nq_small].sum()) / float(np.median(ms_small)) / 1e6}


    # ---- two search contexts on two streams: consecutive small launches overlap (the next one
    # fills the CUs the previous one frees while it drains).  Extra figure, not `value`.
    pipelined = None
    if small is not None:
        ctxs = [pg.SearchContext(ix), pg.SearchContext(ix)]
        streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        Qs = Q[:args.nq_small].contiguous()
        outs = [ix.search_torch(Qs, args.ef), ix.search_torch(Qs, args.ef)]
        torch.cuda.synchronize()
        reps = 12
        for i in range(2):
            ctxs[i].search_torch(Qs, args.ef, outs[i], streams[i])
        torch.cuda.synchronize()
        tp = time.perf_counter()
        for i in range(reps):
            ctxs[i & 1].search_torch(Qs, args.ef, outs[i & 1], streams[i & 1])
        torch.cuda.synchronize()
        tp = time.perf_counter() - tp
        ok = bool((outs[0]["labels"] == labels0[:args.nq_small]).all().item() and
                  (outs[1]["labels"] == labels0[:args.nq_small]).all().item())
        pipelined = {"queries_per_launch": args.nq_small, "launches": reps, "streams": 2,
                     "queries_per_s": args.nq_small * reps / tp, "results_identical": ok}
        for c in ctxs:
            c.close()

    # ---- timed region -----------------------------------------------------------------
    bufs = ix.search_torch(Q, args.ef)           # allocate outputs once
    for _ in range(args.warmup):
        ix.search_torch(Q, args.ef, out=bufs)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ix.search_torch(Q, args.ef, out=bufs)
    barrier()
    elapsed = time.perf_counter() - t0
    same = bool((bufs["labels"] == labels0).all().item())
    # per-launch kernel time of exactly the K timed launches, from the HIP events the library
    # recorded on the launch stream around each kernel
    kernel_ms = [ix.last_search_ms(back) for back in range(min(args.steps, 64))]
    kms = float(np.mean(kernel_ms))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rec_t = torch.tensor([recall], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rec_t, op=dist.ReduceOp.MIN)
    elapsed = float(tmax.item())
    total_queries = args.nq * args.steps * world
    qps = total_queries / elapsed

    achieved = bytes_launch / (kms * 1e-3) / 1e9
    # context for the roof: what a plain device-to-device copy reaches on this GPU (read + write bytes)
    cp_src = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    cp_dst = torch.empty_like(cp_src)
    cp_dst.copy_(cp_src)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        cp_dst.copy_(cp_src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbps = 5 * 2 * cp_src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del cp_src, cp_dst
    result = {
        "metric": "queries/sec at recall@10>=0.95, 1Mx768 L2 efsearch=128",
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"HNSW search: {args.n}x{args.dim} fp32 GMM({args.clusters}, sigma 0.3), "
                        f"{args.metric}, m={args.m}, efconstruction={args.efc} (device build), "
                        f"efsearch={args.ef}, {args.nq} queries/step/GPU resident in HBM",
            "rows": args.n, "dims": args.dim, "m": args.m, "efsearch": args.ef,
            "queries_per_step_per_gpu": args.nq,
            "parallelism": "replica per GPU, queries sharded" if world > 1 else "single GPU",
        },
        "recall_at_10": float(rec_t.item()),
        "results_stable_across_steps": same,
        "evals_per_query": float(E.mean()),
        "hops_per_query": float(H.mean()),
        "alg_bytes_per_query": float(bytes_q.mean()),
        "build_seconds": t_build,
        "datagen_seconds": t_gen,
        "resident_query_slots": ix.last_search_slots(),
        "roofline": {
            "bound": "hbm",
            "kernel": "hnsw_search_kernel",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args),
            "alg_bytes_per_launch": bytes_launch,
            "kernel_ms_per_launch": kms,
            "measured_copy_GBps": copy_gbps,
        },
        "smaller_launch": small,
        "smaller_launch_two_streams": pipelined,
    }

    # ---- CPU baseline: the reference's own code on the same graph bytes, rank 0, N=1 only --
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(args, ix, Q, labels0, func)
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.destroy_process_group()


def main_sharded(args):
    """Row-sharded index (SURVEY.md §8e mode 2, BASELINE config C4): contiguous row ranges, one
    graph per shard, every rank searches the same query batch on its shard, ONE exchange
    (all-gather of (dist,label) lists over RCCL) and the device merge kernel."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch
    from pg_embedding_amd.sharded import ShardedIndex, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]
    lo, hi = shard_range(args.n, world, rank)
    t0 = time.time()
    rows = gmm_torch(hi - lo, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=100 + rank, device=dev)
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    sh = ShardedIndex.build(rows, lo, meta, device=local, max_batch=args.max_batch, ratio=args.ratio)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    del rows
    nq = args.nq
    Q = gmm_torch(nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1, device=dev)   # same on every rank

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        sh.search(Q, args.ef)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        labels, dists, counts = sh.search(Q, args.ef)
    barrier()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ok = bool((counts == args.ef).all().item()) and bool((dists[:, 1:] >= dists[:, :-1]).all().item())
    if rank == 0:
        print(json.dumps({
            "metric": "queries/sec, row-sharded index, per-shard searchKnn + RCCL top-k merge",
            "value": nq * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HNSW search, index of {args.n}x{args.dim} rows sharded over {world} GPU(s) "
                                   f"({hi - lo} rows/shard), {args.metric}, m={args.m}, efsearch={args.ef}, "
                                   f"{nq} queries/step (every GPU searches all of them)",
                       "parallelism": f"row-sharded x{world}, all-gather + merge"},
            "build_seconds": t_build, "merged_results_sorted_and_full": ok}))
    if use_dist:
        dist.destroy_process_group()


def pmc_traffic(args):
    """HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes of THIS command
    ((2*FETCH_SIZE + WRITE_SIZE)*1024, gfx950 FETCH_SIZE correction of the micro-arch guide).  The
    counters cannot be collected from inside the timed process, so the value measured by
    scripts/profile_bench.sh is committed as profiles/traffic.json and reported here only when it
    was taken for the same workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        same = all(t["workload"].get(k) == getattr(args, k) for k in ("n", "dim", "m", "efc", "ef", "nq", "metric"))
        return float(t["hbm_bytes_per_launch"]) if same else None
    except Exception:
        return None


def cpu_baseline(args, ix, Q, gpu_labels, func):
    """oracle/_ref (the unmodified reference distfunc.c + hnswalg.cpp) or, where that was
    not shipped, the C restatement, timed on this box's host cores over a bounded sample of
    the same queries on the identical graph bytes.  Checker/baseline only."""
    import numpy as np
    import oracle

    raw = ix.export_flat()
    kind = "reference" if oracle.have_ref() else "port"
    Cls = oracle.RefIndex if kind == "reference" else oracle.PortIndex
    cpu = Cls(args.dim, args.m, args.efc, args.ef, func, capacity=args.n)
    cpu.load_raw(raw, args.n)
    del raw
    Qh = Q.cpu().numpy()
    ncores = os.cpu_count() or 1
    threads = min(ncores, 64)
    # single thread first (what one Postgres backend gets), sized from a short probe
    probe = cpu.search_many(Qh[:32], args.ef, nthreads=1)
    qps1_est = 32 / max(probe["seconds"], 1e-6)
    n1 = int(max(64, min(args.nq, qps1_est * args.cpu_seconds * 0.4)))
    r1 = cpu.search_many(Qh[:n1], args.ef, nthreads=1)
    qps1 = n1 / r1["seconds"]
    nt = int(max(threads * 8, min(args.nq, qps1 * threads * args.cpu_seconds * 0.6)))
    rt = cpu.search_many(Qh[:nt], args.ef, nthreads=threads)
    qpst = nt / rt["seconds"]
    # 8 threads: the figure BASELINE.md §3 plans next to the single-thread one
    n8 = int(max(64, min(args.nq, qps1 * 8 * args.cpu_seconds * 0.25)))
    r8 = cpu.search_many(Qh[:n8], args.ef, nthreads=min(8, ncores))
    qps8 = n8 / r8["seconds"]
    glab = gpu_labels[:n1].cpu().numpy().view(np.uint64)
    agree = float((r1["labels"] == glab).all(axis=1).mean())
    return {
        "value": qpst, "unit": "queries/s", "cores": threads, "kind": kind,
        "sample": f"{nt} of the {args.nq} queries on {threads} host threads (one query per thread, "
                  f"shared read-only index); single thread: {n1} queries",
        "single_thread_qps": qps1,
        "eight_thread_qps": qps8,
        "host_cpus": ncores,
        "fraction_of_queries_with_identical_ids": agree,
    }


if __name__ == "__main__":
    main()
