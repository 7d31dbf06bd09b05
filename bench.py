#!/usr/bin/env python3
"""bench.py — queries/sec of the HNSW search hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]

A "step" is one pass of the hot path over one batch of `--nq` (default 40 000) synthetic queries
(already resident in HBM): the fused searchBaseLayer/searchKnn kernel over an HBM-resident index.
Workload at N=1 = the configuration the metric is quoted on: 1M x 768 fp32, L2,
efsearch=128 — on the graph the REFERENCE itself builds for the table (oracle/_ref's serial hnsw_bind_point, link words uploaded byte
for byte) where that travels with the tree, otherwise on a graph built in HBM by the device insert path before the timed region.

N>1: one process per GPU over RCCL (torch.distributed backend "nccl").  Started by the driver under
`python -m torch.distributed.run ...` it reads RANK/LOCAL_RANK/WORLD_SIZE; started plainly as
`python bench.py --gpus N` it launches those N ranks ITSELF (re-exec under torch.distributed.run,
rendezvous on 127.0.0.1).  --mode replicas (default, the headline metric): every rank holds a replica of
the index and its own query batch (queries are the independent units; no data-path collective) -> weak
scaling.  --mode sharded (BASELINE config C4): the rows are partitioned over the ranks, every rank
searches every query on its shard, ONE packed all-gather + the device merge kernel -> strong scaling.

Prints ONE JSON line on rank 0 (see the contract in the task description) with two extra
objects: "roofline" (dominant kernel vs the HBM roof from HIP events on the kernel's own stream: algorithmic bytes over
the nominal peak, the REPLAY of the launch's own row trace as the roof the kernel cannot beat, the share of its reads no
cache can hold, and the same on a cache-hostile table: roofline.hbm_only) and "cpu_baseline" (the reference's CPU path on the same graph bytes, bounded sample,
host cores of this box, with every id mismatch against the reference classified).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from pg_embedding_amd import watchdog                      # noqa: E402
watchdog.arm(default_seconds=3000.0, env_sync=False)                    # --timeout SECONDS: a hung launch ends the run with status 124, not never

# The two-stream legs keep launches in flight on two HIP streams.  HIP spreads streams over GPU_MAX_HW_QUEUES hardware queues (4 by
# default) and launches that share a queue run one after the other — the narrow-row pair showed no overlap at all in the bench process
# (0.60 of nominal) and full overlap in a fresh process (0.70, profiles/r6d_*): which queues two streams get depends on how many the
# process has created before.  More queues than the bench ever creates streams; read by the runtime at its first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
REL_TOL = 1e-5            # north-star tolerance (BASELINE.json)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", dest="n", type=int, default=0, help="index rows (default 1M; 10M in --mode sharded)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--hnsw-m", dest="m", type=int, default=16, help="m reloption (maxM = 2m)")
    ap.add_argument("--efc", type=int, default=200, help="efconstruction for the device build")
    ap.add_argument("--ef", type=int, default=128, help="efsearch")
    ap.add_argument("--nq", type=int, default=0, help="queries per step per GPU (default 40000; 1024 in --mode sharded)")
    ap.add_argument("--nq-small", type=int, default=10_000, help="also report a smaller launch (0 = skip)")
    ap.add_argument("--metric", default="l2", choices=["l2", "cosine", "manhattan"])
    ap.add_argument("--clusters", type=int, default=1000)
    ap.add_argument("--max-batch", type=int, default=0)
    ap.add_argument("--ratio", type=int, default=0)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample time")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--recall-queries", type=int, default=1000)
    ap.add_argument("--alloc-probe", action="store_true",
                    help="diagnostic: do not empty torch's cache in front of the side configs and time the first one again with re-created tensors (profiles/r5af_*)")
    ap.add_argument("--profile-config", default="", choices=["", "M", "C2", "C3", "C5", "hostile"],
                    help="build and search ONE configuration as the bench does (its --steps timed launches and nothing else): the command "
                         "scripts/profile_configs.sh runs under rocprofv3")
    ap.add_argument("--device-build", action="store_true",
                    help="headline on the batched device build of torch-generated rows even where the reference's own graph of the table "
                         "(oracle/_ref/serial_graph_*.npy) is present")
    ap.add_argument("--no-side-configs", action="store_true",
                    help="skip the other BASELINE configs and the stress datasets (reported extras, N=1 only)")
    ap.add_argument("--hostile-rows", type=int, default=8_000_000,
                    help="rows of the second timed dataset (roofline.hbm_only): clusters of 200 rows, every query of a launch from a "
                         "cluster of its own, far larger than the caches; 0 = skip; skipped for N>1")
    ap.add_argument("--serial-rows", type=int, default=40_000,
                    help="the serial-vs-batched build comparison (the reference's serial hnsw_bind_point order against the batched device "
                         "build the headline index uses, same rows, same queries): rows of the device-built fallback, used where the "
                         "reference's own serial graph of the headline table (oracle/_ref/serial_graph_*.npy) does not travel with the "
                         "tree; 0 = skip the leg; N=1 only")
    ap.add_argument("--shards", type=int, default=0, help="--mode sharded-native: row shards (default one per device)")
    ap.add_argument("--hostile-m", type=int, default=32, help="m of the cache-hostile table (the recall gate must hold there too)")
    ap.add_argument("--mode", default="replicas", choices=["replicas", "sharded", "sharded-native"],
                    help="replicas: index mirrored on every GPU, queries sharded (headline metric). sharded: rows partitioned "
                         "across GPUs (config C4), every GPU searches every query, one packed RCCL all-gather + merge kernel. "
                         "sharded-native: the same partition inside ONE process (hnsw_gpu_sharded_search_dev: shards on --gpus "
                         "devices, result lists stored into the merging device over xGMI peer access, one merge kernel)")
    ap.add_argument("--no-multi-gpu-extras", action="store_true",
                    help="N>1 replicas runs also time both row-sharded hosts on a small index after the headline region "
                         "(reported under multi_gpu_extras, never part of `value`); this skips them")
    a = ap.parse_args()
    if a.n == 0:
        a.n = 10_000_000 if a.mode.startswith("sharded") else 1_000_000
    if a.nq == 0:
        a.nq = 1024 if a.mode.startswith("sharded") else 40_000
    return a


# ------------------------------------------------------------------------------------------ launcher
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args) -> int:
    """`python bench.py --gpus N` with no rendezvous in the environment: start the N ranks here, one per GPU,
    exactly as the driver would (torch.distributed.run, 127.0.0.1)."""
    selftest = os.environ.get("PGEMB_BENCH_SELFTEST") == "1"
    if not selftest:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            print(f"bench.py: --gpus {args.gpus} but only {have} device(s) visible", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    env["PGEMB_BENCH_SELF_LAUNCHED"] = "1"
    return subprocess.run(cmd, env=env).returncode


def init_ranks(args):
    """(world, rank, local, use_dist, backend) after joining the process group when there is one."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ       # launched by torch.distributed.run
    backend = None
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if os.environ.get("PGEMB_BENCH_SELFTEST") == "1":
            dist.init_process_group("gloo")
        else:
            import datetime
            torch.cuda.set_device(local)
            # (a rank that dies must cost the others minutes, not the default half hour)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=300))
        backend = dist.get_backend()
        world = dist.get_world_size()                                   # what the collective library reports
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but {world} rank(s) joined", file=sys.stderr)
    return world, rank, local, use_dist, backend


def selftest_main(args):
    """CPU check of the launch path (tests/test_bench_launch.py): ranks rendezvous over gloo, agree on the
    world size through a collective, rank 0 prints one line.  No device work.  --mode sharded also runs the row-sharded
    layout's exchange step for real — pg_embedding_amd.sharded.ShardedIndex with a synthetic per-rank search and a torch merge
    injected: the packed block, the ONE all-gather per search, the persistent buffers, the per-rank step breakdown the device
    run reports — so that the first N>1 run on hardware is not the first run of that code."""
    import torch
    import torch.distributed as dist
    world, rank, local, use_dist, backend = init_ranks(args)
    t = torch.ones(1)
    if use_dist:
        dist.all_reduce(t)
    line = {"selftest": True, "n_gpus": world, "ranks_joined": int(t.item()), "backend": backend,
            "mode": args.mode, "self_launched": os.environ.get("PGEMB_BENCH_SELF_LAUNCHED") == "1"}
    if args.mode == "sharded":
        from pg_embedding_amd.sharded import ShardedIndex, block_bytes
        nq, ef, steps = 64, 16, max(1, args.steps)

        def shard_lists(r, step):                               # rank r's result lists: ascending (dist, label), labels unique across ranks
            g = torch.Generator().manual_seed(1000 * step + r)
            d, _ = torch.sort(torch.rand((nq, ef), generator=g), dim=1)
            lab = torch.arange(nq * ef, dtype=torch.int64).reshape(nq, ef) * world + r
            return lab, d.to(torch.float32)

        def merge(lab, dst, k):                                 # [world, nq, ef] -> the k best by (dist, label)
            L = lab.permute(1, 0, 2).reshape(nq, -1)
            D = dst.permute(1, 0, 2).reshape(nq, -1)
            order = torch.argsort(D * 1.0, dim=1, stable=True)
            return torch.gather(L, 1, order)[:, :k], torch.gather(D, 1, order)[:, :k], torch.full((nq,), k, dtype=torch.int32)

        step_no = [0]
        sh = ShardedIndex(local_search=lambda q, k: shard_lists(rank, step_no[0]), merge=merge)
        sh.record_timing = True
        ok = True
        for s_ in range(steps):
            step_no[0] = s_
            lab, dst, cnt = sh.search(torch.zeros((nq, 4)), ef)
            every = [shard_lists(r, s_) for r in range(world)]
            wl, wd, _ = merge(torch.stack([e[0] for e in every]), torch.stack([e[1] for e in every]), ef)
            ok = ok and bool((lab == wl).all()) and bool((dst == wd).all())
        mine = torch.tensor([[float(sum(x[k] for x in sh.timings_ms()) / steps) for k in range(3)] + [1.0 if ok else 0.0]], dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        if use_dist:
            dist.all_gather(every, mine)
        else:
            every = [mine]
        line.update({"exchange": {"collectives_per_step": sh.exchanges / steps, "bytes_per_rank": block_bytes(nq, ef)},
                     "buffers_allocated": len(sh._bufs),
                     "merged_equals_the_global_order_on_every_rank": all(float(e[0][3]) == 1.0 for e in every),
                     "step_breakdown_ms_per_rank": [{"local_search_ms": float(e[0][0]), "exchange_ms": float(e[0][1]), "merge_ms": float(e[0][2])} for e in every]})
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ helpers
def alg_bytes(stats, counts, dim, m):
    """algorithmic bytes per query, SURVEY.md §8(d): B_q = E_q*dim*4 + H_q*(maxM+1)*4 + dim*4 + R_q*8"""
    E, H = stats[:, 0], stats[:, 1]
    return E * dim * 4 + H * (2 * m + 1) * 4 + dim * 4 + counts * 8


def random_gather(ix):
    """best dependency-free gather rate of RANDOM whole rows of this mirror's row table (csrc/device_roof.h).  Informational
    only: random rows share nothing, a launch's rows do (the queries of a launch cross the same hubs and clusters), so this is
    NOT a roof of the search kernel — the replay of the launch's own trace (trace_roof) is."""
    best, cfg = 0.0, None
    for wpc in (8, 16):
        for t in (8, 12, 16, 24):
            g = ix.gather_roof(t, wpc, 200)
            if g > best:
                best, cfg = g, {"loads_per_lane": t, "waves_per_cu": wpc}
    return best, cfg


MALL_BYTES = 256 << 20        # Infinity Cache of one MI355X (memory-side)
L2_BYTES = 8 * (4 << 20)      # 8 XCDs x 4 MB


def trace_roof(ix, Q, ef):
    """Replay roof + reuse analysis of ONE launch's own evaluation trace (include/hnsw_gpu.h, hnsw_gpu_search_traced_dev /
    hnsw_gpu_replay_roof).

    replay: the rows the launch scored, gathered again by the same number of resident waves in the same query order with the
      search kernel's load shape and NOTHING in between (no pop, link list, visited test, accept loop).  Same bytes, same
      locality between concurrent queries, no dependent chain: kernel_ms / replay_ms is what the chain costs (<= 1 by
      construction), replay GB/s is what the memory system gives this trace.
    reuse: every row read gets a time (its query's start/end device-clock stamps, spread evenly over the query's reads); a
      read is `far` when the same row was not read within the last 256 MB of the launch's row traffic (first touch
      included): no cache on the chip can hold it, so it comes from HBM.  `hbm_lower_bound_GBps` = far bytes / kernel time:
      what HBM provably delivered; the rest of `achieved` MAY be cache service."""
    import torch
    nq = Q.shape[0]
    cap = 4096
    tr = ix.search_traced_torch(Q, ef, evals_cap=cap)
    torch.cuda.synchronize()
    E = tr["stats"][:, 0].to(torch.int64)
    if int(E.max().item()) > cap:                            # rare: long walks; one retry with room for all of them
        cap = int(E.max().item()) + 64
        tr = ix.search_traced_torch(Q, ef, evals_cap=cap)
        torch.cuda.synchronize()
        E = tr["stats"][:, 0].to(torch.int64)
    slots = ix.last_search_slots()
    row_bytes = int(ix.meta.dim + 3) // 4 * 16
    best = None
    lpr = (row_bytes // 16 + 15) // 16                        # 16-byte loads per lane per row
    shapes = [(2, 2), (2, 4), (2, 8)] if lpr <= 2 else [(4, 2), (4, 4)] if lpr <= 4 else [(8, 2), (4, 4)] if lpr <= 8 else [(12, 2), (12, 1), (6, 4)]
    for kb, rpg in shapes:                                    # a roof: the best of the search kernel's own shape and its neighbours ...
        for rs in (slots, 2 * slots):                         # ... at the search's own occupancy and with twice as many waves
            ms, by = ix.replay_roof(tr, rs, kb, rpg)
            if best is None or ms < best[0]:
                best = (ms, by, f"<{kb},{rpg}>", rs)
    rms, rbytes, rlpl, rslots = best
    out = {"replay_ms": rms, "replay_row_bytes": rbytes, "replay_GBps": rbytes / rms / 1e6, "replay_shape": rlpl,
           "replay_slots": rslots, "search_slots": slots}
    # ---- reuse distances, on the device with torch (tens of millions of reads) ----
    j = torch.arange(cap, device=Q.device, dtype=torch.int64)[None, :]
    valid = j < E[:, None]
    t0 = tr["times"][:, 0].to(torch.float64)[:, None]
    t1 = tr["times"][:, 1].to(torch.float64)[:, None]
    t = (t0 + (t1 - t0) * (j.to(torch.float64) + 0.5) / E.clamp(min=1)[:, None].to(torch.float64))[valid]
    rows = tr["evals"].to(torch.int64)[valid]
    del valid, j
    nreads = rows.numel()
    order = torch.argsort(t)                                  # time order of all reads of the launch
    rank = torch.empty_like(order)
    rank[order] = torch.arange(nreads, device=Q.device)
    del order, t
    key, _ = torch.sort(rows * (1 << 32) + rank)              # by row, then by time
    r_sorted, k_sorted = key >> 32, key & 0xFFFFFFFF
    same = torch.zeros(nreads, dtype=torch.bool, device=Q.device)
    same[1:] = r_sorted[1:] == r_sorted[:-1]
    gap = torch.full((nreads,), 1 << 40, dtype=torch.int64, device=Q.device)
    gap[1:] = torch.where(same[1:], (k_sorted[1:] - k_sorted[:-1]) * row_bytes, gap[1:])
    distinct = int((~same).sum().item())
    far_mall = float((gap > MALL_BYTES).double().mean().item())
    far_l2 = float((gap > L2_BYTES).double().mean().item())
    out.update({"row_reads": nreads, "distinct_rows": distinct, "reads_per_distinct_row": nreads / max(distinct, 1),
                "reads_beyond_infinity_cache_reach": far_mall, "reads_beyond_l2_reach": far_l2, "_far_bytes": far_mall * nreads * row_bytes,
                "traced_launch_clock_span_ms": float((tr["times"][:, 1].max() - tr["times"][:, 0].min()).item()) / 1e5})
    del tr
    return out


def finish_trace_roof(tr, kernel_ms, alg_bytes_launch):
    """ratios of an (untraced) launch of `kernel_ms` against its trace's replay and reuse figures"""
    tr = dict(tr)
    far_bytes = tr.pop("_far_bytes")
    tr["replay_ms_over_kernel_ms"] = tr["replay_ms"] / kernel_ms                       # <= 1: the rest is the dependent chain
    tr["frac_of_replay"] = (alg_bytes_launch / kernel_ms) / (tr["replay_row_bytes"] / tr["replay_ms"])
    tr["hbm_lower_bound_GBps"] = far_bytes / kernel_ms / 1e6
    return tr


def copy_roof(dev):
    import torch
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
    return 5 * 2 * cp_src.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


def build_index(args, n, clusters, dev, local, func):
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch
    t0 = time.time()
    X = gmm_torch(n, args.dim, k=clusters, sigma=0.3, seed=42, device=dev)
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    ix = pg.GpuIndex.empty(meta, n, device=local)
    ix.append_torch(X)
    torch.cuda.synchronize()
    t_gen = time.time() - t0
    t0 = time.time()
    ix.link(0, n, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    del X
    return ix, t_gen, t_build


def reference_graph_path(args, func):
    """the reference's own serial graph of the headline table, if it travels with the tree (tests/experiments/make_ref_serial_graph.py)"""
    path = REF_GRAPH.format(n=args.n, dim=args.dim, m=args.m, efc=args.efc)
    ok = func == 0 and args.clusters == 1000 and not args.device_build and os.path.exists(path)
    return path if ok else None


def build_index_on_reference_graph(args, path, dev, local, func):
    """The index the metric is about, as the REFERENCE builds it (hnswalg.cpp:279-291: serial hnsw_bind_point): link words made once by
    oracle/_ref on a host core, rows from the seeded numpy generator (the bytes that graph was built over), uploaded as element images
    chunk by chunk — the host never holds the 3 GB table.  Returns (index, upload seconds, figures of the graph)."""
    import numpy as np
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_chunks
    links = np.load(path, mmap_mode="r")
    n = links.shape[0]
    assert n == args.n and links.shape[1] == 2 * args.m + 1
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    es, od, ol = int(meta.size_data_per_element), int(meta.offset_data), int(meta.offset_label)
    import torch
    ix = pg.GpuIndex.empty(meta, n, device=local)
    t0 = time.time()
    # (the reference's link lists point anywhere in the table, and an element image is validated against the elements the mirror
    # holds: first n unlinked zero rows, then the images replace them chunk by chunk)
    z = torch.zeros((1 << 16, args.dim), dtype=torch.float32, device=dev)
    for a in range(0, n, 1 << 16):
        ix.append_torch(z[:min(1 << 16, n - a)])
    torch.cuda.synchronize()
    del z
    for a, x in gmm_chunks(n, args.dim, k=args.clusters, sigma=0.3, seed=42, chunk=1 << 16):
        b = a + x.shape[0]
        raw = np.zeros((b - a, es), np.uint8)
        raw[:, :od] = np.ascontiguousarray(links[a:b]).view(np.uint8)
        raw[:, od:ol] = x.view(np.uint8)
        raw[:, ol:] = np.arange(a, b, dtype=np.uint64)[:, None].view(np.uint8)
        ix.update_from_flat(raw.reshape(-1), a, b - a)
    deg = np.asarray(links[:, 0])
    return ix, time.time() - t0, {"mean_degree": float(deg.mean()), "full_lists": float((deg == 2 * args.m).mean())}


# ------------------------------------------------------------------------------------------ replicas
def main():
    args = parse()
    if args.mode == "sharded-native" and os.environ.get("PGEMB_BENCH_SELFTEST") != "1":
        return main_sharded_native(args)                     # one process whatever --gpus says
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    if os.environ.get("PGEMB_BENCH_SELFTEST") == "1":
        return selftest_main(args)
    if args.mode == "sharded":
        return main_sharded(args)
    if args.profile_config:
        return profile_config_main(args)
    import numpy as np
    import torch
    import torch.distributed as dist

    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k

    world, rank, local, use_dist, backend = init_ranks(args)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- index.  Where the reference's OWN graph of the headline table travels with the tree (oracle/_ref/serial_graph_*: serial
    # hnsw_bind_point by the unmodified reference, hnswalg.cpp:279-291) `value` is measured on it, uploaded byte for byte — the graph
    # the metric is about; the batched device build of the same rows is then the side leg (serial_vs_batched_build).  Otherwise:
    # synthetic rows generated in HBM, graph built by the device insert path.
    ref_path = reference_graph_path(args, func)
    ref_graph = None
    if ref_path:
        ix, t_gen, ref_graph = build_index_on_reference_graph(args, ref_path, dev, local, func)
        t_build = None
        from pg_embedding_amd.datasets import gmm
        Q = torch.from_numpy(gmm(args.nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1 + rank)).to(dev)     # (the numpy generator's mixture)
        built_by = ("graph built by the REFERENCE itself (oracle/_ref: unmodified hnswalg.cpp, serial hnsw_bind_point on a host core; "
                    "link words uploaded byte for byte)")
    else:
        ix, t_gen, t_build = build_index(args, args.n, args.clusters, dev, local, func)
        # every rank searches its own query stream (weak scaling)
        Q = gmm_torch(args.nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1 + rank, device=dev)
        built_by = "device build"

    # ---- recall@10 against exhaustive search with the same metric ---------------------
    nrec = min(args.recall_queries, args.nq)
    truth, _ = ix.bruteforce_torch(Q[:nrec].contiguous(), 10, mfma=True)
    out = ix.search_torch(Q, args.ef, stats=True)
    torch.cuda.synchronize()
    labels0 = out["labels"].clone()
    recall = recall_at_k(labels0[:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
    stats = out["stats"].cpu().numpy().astype(np.int64)
    counts = out["counts"].cpu().numpy().astype(np.int64)
    E, H = stats[:, 0], stats[:, 1]
    bytes_q = alg_bytes(stats, counts, args.dim, args.m)
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
        pipelined = two_streams(ix, Q[:args.nq_small].contiguous(), args.ef, labels0[:args.nq_small], dev)

    # ---- single query per launch: the reference's own call shape (embedding.c:317), device time only.  Measured twice: right here,
    # behind seconds of sustained full-chip load (the device's clocks are down: the state `value` is measured in), and after one
    # second of idle (the state a backend that calls once in a while finds the device in) — the same 40 queries, medians of both.
    single = None
    if rank == 0:
        q1 = Q[:1].contiguous()
        b1 = ix.search_torch(q1, args.ef)

        def one_by_one():
            ms1 = []
            for i in range(48):
                ix.search_torch(Q[i:i + 1].contiguous(), args.ef, out=b1)
                ms1.append(ix.last_search_ms())
            return ms1
        hot = one_by_one()
        torch.cuda.synchronize()
        time.sleep(1.0)
        idle = one_by_one()
        single = {"kernel_ms_median": float(np.median(hot[8:])), "kernel_ms_mean": float(np.mean(hot[8:])),
                  "kernel_ms_median_after_idle": float(np.median(idle[8:])), "kernel": ix.last_search_kernel()}

    # ---- roofs measured on this device: dependency-free gather of this table's rows, plain copy
    g_rand, g_cfg = random_gather(ix)
    copy_gbps = copy_roof(dev)
    # ---- this launch's own trace: replay roof + reuse distances (BEFORE the timed region: the traced launch stores 6 KB
    # more per query and must not be one of the launches the timing and profiles/ summarise)
    tr_raw = trace_roof(ix, Q, args.ef) if rank == 0 else None

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
    kernel_name = ix.last_search_kernel()
    # per-launch kernel time of exactly the K timed launches, from the HIP events the library
    # recorded on the launch stream around each kernel
    kernel_ms = [ix.last_search_ms(back) for back in range(min(args.steps, 64))]
    kms = float(np.mean(kernel_ms))

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    rec_t = torch.tensor([recall], dtype=torch.float64, device=dev)
    per_rank = [elapsed]
    per_rank_kms = [kms]
    per_rank_bytes = [bytes_launch]
    if use_dist:
        mine = torch.tensor([elapsed, kms, bytes_launch], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(t[0].item()) for t in every]
        per_rank_kms = [float(t[1].item()) for t in every]
        per_rank_bytes = [float(t[2].item()) for t in every]
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(rec_t, op=dist.ReduceOp.MIN)
    elapsed = float(tmax.item())
    total_queries = args.nq * args.steps * world
    qps = total_queries / elapsed

    achieved = bytes_launch / (kms * 1e-3) / 1e9
    traffic, traffic_src = pmc_traffic(args, kernel_name, "reference" if ref_path else "device (batched)")
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
                        f"{args.metric}, m={args.m}, efconstruction={args.efc} ({built_by}), "
                        f"efsearch={args.ef}, {args.nq} queries/step/GPU resident in HBM",
            "graph_built_by": "reference" if ref_path else "device (batched)",
            "rows": args.n, "dims": args.dim, "m": args.m, "efsearch": args.ef,
            "queries_per_step_per_gpu": args.nq,
            "parallelism": "replica per GPU, queries sharded" if world > 1 else "single GPU",
        },
        "ranks": {"world_size_reported_by_backend": world, "backend": backend or "none (single process)",
                  "self_launched": os.environ.get("PGEMB_BENCH_SELF_LAUNCHED") == "1",
                  "queries_per_s_per_rank": [args.nq * args.steps / t for t in per_rank],
                  # every rank's own dominant-kernel figures (its own queries, its own HIP events): the 1 -> N curve rank by rank
                  "kernel_ms_per_launch_per_rank": per_rank_kms,
                  "roofline_frac_per_rank": [b / (k * 1e-3) / 1e9 / HBM_PEAK_GBS for b, k in zip(per_rank_bytes, per_rank_kms)]},
        "recall_at_10": float(rec_t.item()),
        "results_stable_across_steps": same,
        "shader_clock_mhz": ix.last_search_clock_mhz(),
        "evals_per_query": float(E.mean()),
        "hops_per_query": float(H.mean()),
        "alg_bytes_per_query": float(bytes_q.mean()),
        "build_seconds": t_build,
        "datagen_seconds": t_gen,
        "resident_query_slots": ix.last_search_slots(),
        "roofline": {
            "bound": "hbm",
            "kernel": kernel_name,
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_src,
            "alg_bytes_per_launch": bytes_launch,
            "kernel_ms_per_launch": kms,
            # `achieved` is ALGORITHMIC bytes over kernel time: rows that queries of one launch share may be served by L2 /
            # Infinity Cache.  `replay` = the launch's own row trace gathered again with nothing in between (the kernel
            # cannot beat it: frac_of_replay <= 1) + how many of its reads no cache can hold (hbm_lower_bound_GBps);
            # `hbm_only` below = the same figures on the cache-hostile table.
            "replay": finish_trace_roof(tr_raw, kms, bytes_launch) if tr_raw else None,
            "random_row_gather_GBps": g_rand,
            "random_row_gather_config": g_cfg,
            "measured_copy_GBps": copy_gbps,
        },
        "smaller_launch": small,
        "smaller_launch_two_streams": pipelined,
        "single_query_launch": single,
    }

    # ---- the host-pointer form of the same batch (hnsw_gpu_search_batch): what a caller without device buffers pays on top
    # (Everything from here on is an extra leg next to a headline that is already complete: a leg that fails says so under its own key
    # — with the exception's text — and the line is printed all the same.)
    def leg(fn, *fargs):
        try:
            return fn(*fargs)
        except Exception as e:                                   # noqa: BLE001 (reported, not swallowed)
            import traceback
            sys.stderr.write(traceback.format_exc())
            return {"failed": f"{type(e).__name__}: {e}"[:400]}

    if rank == 0 and world == 1:
        result["host_pointer_batch"] = leg(host_pointer_batch, args, ix, Q, labels0)
    # ---- CPU baseline: the reference's own code on the same graph bytes, rank 0, N=1 only --
    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = leg(cpu_baseline, args, ix, Q, labels0, out["dists"], func)
    ix.close()
    del ix, out, bufs, labels0

    # ---- a dataset whose rows do not repeat inside a launch (N=1 only): the Infinity-Cache share made visible
    if rank == 0 and world == 1 and args.hostile_rows > 0:
        result["roofline"]["hbm_only"] = leg(hostile, args, dev, local, func)
    # ---- the other BASELINE configs and the stress datasets of SURVEY.md §8(d), same kernels, N=1 only (extras, not `value`)
    if rank == 0 and world == 1 and args.serial_rows > 0:
        head_serial = None
        if ref_path:                                            # the headline WAS the reference's graph: its figures are the leg's "serial" side
            head_serial = dict(ref_graph, evals_per_query=float(E.mean()), hops_per_query=float(H.mean()), recall_at_10=recall,
                               alg_bytes_per_query=float(bytes_q.mean()), queries_per_s=args.nq / kms * 1e3, kernel_ms_per_launch=kms)
        result["serial_vs_batched_build"] = leg(serial_vs_batched, args, dev, local, func, min(args.serial_rows, args.n), 10_000, head_serial)
    if rank == 0 and world == 1 and not args.no_side_configs:
        result["other_configs"] = leg(side_configs, args, dev, local)
        result["serial_insert"] = leg(serial_insert, args, dev)
    if use_dist and world > 1 and not args.no_multi_gpu_extras:
        multi_gpu_extras(args, result, world, rank, local, dev)       # prints the lines itself (also when the extras time out)
    elif rank == 0:
        emit(result)
    if use_dist:
        dist.destroy_process_group()


HEADLINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline", "cpu_baseline")


def emit(result):
    """ONE JSON line on stdout — the line the contract describes, compact enough that the tail of a captured stdout holds it whole — and,
    BEFORE it, every extra leg as one long JSON line (`"extras_of": <metric>`) on stderr (also written to gpurun_out/bench_extras.json):
    whoever captures the two streams together sees the extras first and the headline last, whoever parses stdout finds exactly one
    line.  The headline carries, as plain scalars inside `config` / `roofline` / `cpu_baseline` (the objects a record keeps),
    everything the metric's claim rests on: the recall gate, the one-query latency, the HBM-only fraction, traffic over algorithmic
    bytes, and the other configs' fractions."""
    def get(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    head = {k: result[k] for k in HEADLINE_KEYS if k in result}
    cfg = dict(head["config"])
    cfg["recall_at_10"] = result.get("recall_at_10")
    cfg["recall_gate_0.95_holds"] = bool(result.get("recall_at_10", 0.0) >= 0.95)
    cfg["evals_per_query"] = result.get("evals_per_query")
    cfg["hops_per_query"] = result.get("hops_per_query")
    cfg["results_stable_across_steps"] = result.get("results_stable_across_steps")
    cfg["shader_clock_mhz"] = result.get("shader_clock_mhz")
    svb = result.get("serial_vs_batched_build")
    if isinstance(svb, dict) and "serial" in svb:
        cfg["serial_vs_batched_rows"] = svb["rows"]
        cfg["serial_graph_built_by"] = svb.get("serial_graph_built_by")
        cfg["serial_build_evals_per_query"] = svb["serial"]["evals_per_query"]
        cfg["batched_build_evals_per_query"] = svb["batched"]["evals_per_query"]
        cfg["serial_build_recall_at_10"] = svb["serial"]["recall_at_10"]
        cfg["batched_build_recall_at_10"] = svb["batched"]["recall_at_10"]
        cfg["batched_build_within_2_percent_of_serial"] = svb["within_2_percent"]
    head["config"] = cfg
    roof = {k: v for k, v in head["roofline"].items() if not isinstance(v, (dict, list))}
    roof["traffic_over_algorithmic"] = (roof["traffic"] / roof["alg_bytes_per_launch"]) if roof.get("traffic") else None
    roof["frac_of_replay"] = get(result, "roofline", "replay", "frac_of_replay")
    roof["hbm_lower_bound_GBps"] = get(result, "roofline", "replay", "hbm_lower_bound_GBps")
    roof["hbm_only_frac"] = get(result, "roofline", "hbm_only", "frac")
    roof["hbm_only_algorithmic_frac"] = get(result, "roofline", "hbm_only", "algorithmic_frac")
    roof["hbm_only_recall_at_10"] = get(result, "roofline", "hbm_only", "recall_at_10")
    roof["single_query_ms"] = get(result, "single_query_launch", "kernel_ms_median")
    roof["single_query_ms_after_idle"] = get(result, "single_query_launch", "kernel_ms_median_after_idle")
    for key, name in (("C2_sift_like_1Mx128_l2_m16", "c2"), ("C3_1Mx768_cosine_m32", "c3"), ("C5_1Mx1536_cosine_m32_Q1024", "c5")):
        roof[name + "_frac"] = get(result, "other_configs", key, "frac_of_8TBps")
        roof[name + "_recall_at_10"] = get(result, "other_configs", key, "recall_at_10")
        roof[name + "_traffic_over_algorithmic"] = get(result, "other_configs", key, "traffic_over_algorithmic")
    # the narrow-row launch: one state or two?  (profiles/r5af_*) — spread of its 12 timed launches, the clock it ran at, where its mirror sits
    mmm = get(result, "other_configs", "C2_sift_like_1Mx128_l2_m16", "kernel_ms_min_median_max")
    if mmm:
        roof["c2_kernel_ms_min"], roof["c2_kernel_ms_median"], roof["c2_kernel_ms_max"] = mmm
    roof["c2_shader_clock_mhz"] = get(result, "other_configs", "C2_sift_like_1Mx128_l2_m16", "shader_clock_mhz")
    roof["c2_mirror_arrays_on_2MiB_boundaries"] = get(result, "other_configs", "C2_sift_like_1Mx128_l2_m16", "mirror_arrays_on_2MiB_boundaries")
    roof["c2_frac_two_streams"] = get(result, "other_configs", "C2_sift_like_1Mx128_l2_m16", "two_streams", "frac_of_8TBps")
    roof["hbm_only_traffic_over_algorithmic"] = get(result, "roofline", "hbm_only", "traffic_over_algorithmic")
    roof["c5_kernel_ms"] = get(result, "other_configs", "C5_1Mx1536_cosine_m32_Q1024", "kernel_ms_per_launch")
    roof["mfma_gemm_frac"] = get(result, "other_configs", "C5_1Mx1536_cosine_m32_Q1024", "exhaustive_mfma_gemm", "frac")
    roof["mfma_gemm_tflops"] = get(result, "other_configs", "C5_1Mx1536_cosine_m32_Q1024", "exhaustive_mfma_gemm", "tflops")
    roof["insert_one_ms"] = get(result, "serial_insert", "insert_one_ms_median")
    head["roofline"] = roof
    if isinstance(head.get("cpu_baseline"), dict):
        head["cpu_baseline"] = {k: v for k, v in head["cpu_baseline"].items() if not isinstance(v, (dict, list))}
    extras = {"extras_of": result.get("metric"), "note": "every extra leg of this run; the headline line follows as the LAST line"}
    extras.update({k: v for k, v in result.items() if k not in ("metric", "value", "unit")})
    line = json.dumps(extras)
    sys.stderr.write(line + "\n")
    sys.stderr.flush()
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "w") as f:
            f.write(line + "\n" + json.dumps(head) + "\n")
    except OSError:
        pass
    sys.stdout.write(json.dumps(head) + "\n")
    sys.stdout.flush()


def multi_gpu_extras(args, result, world, rank, local, dev):
    """After the headline region of an N>1 replicas run: both hosts of the ROW-SHARDED layout (SURVEY.md 8e mode 2) on a small
    index, so that a node with several GPUs measures them whenever it measures the headline — (a) one process per GPU, one packed
    all-gather over RCCL + merge (pg_embedding_amd/sharded.py), every rank takes part; (b) one process, shards on all N devices,
    peer stores + merge (hnsw_gpu_sharded_search_dev), run by rank 0 in a child process while the others wait.  Extras only:
    `value` is complete before this starts, every failure is caught and reported as text, and a timer prints the headline line
    and ends the process should any of it hang — the headline never depends on code that has not run on this node before."""
    import threading
    import numpy as np
    import torch
    import torch.distributed as dist
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch
    done = threading.Event()
    limit = 240.0

    def expire():
        if not done.wait(limit):
            if rank == 0:
                result["multi_gpu_extras"] = {"error": f"did not finish within {limit:.0f} s; the headline figures above are complete"}
                emit(result)
            os._exit(0)
    threading.Thread(target=expire, daemon=True).start()
    extras = {}
    # ---- (a) row-sharded over RCCL: 250 000 rows per rank, 1024 queries, 5 steps
    try:
        from pg_embedding_amd.sharded import ShardedIndex, block_bytes
        rows_per = 250_000
        func = pg.DIST_L2
        lo = rows_per * rank
        rows = gmm_torch(rows_per, args.dim, k=1000, sigma=0.3, seed=42, stream=200 + rank, device=dev)
        meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
        sh = ShardedIndex.build(rows, lo, meta, device=local, max_batch=args.max_batch, ratio=args.ratio)
        del rows
        Qx = gmm_torch(1024, args.dim, k=1000, sigma=0.3, seed=42, stream=1, device=dev)
        sh.search(Qx, args.ef)
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        sh.record_timing = True
        t0 = time.perf_counter()
        for _ in range(5):
            labels, dists, counts = sh.search(Qx, args.ef)
        torch.cuda.synchronize(); dist.barrier()
        el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        st = sh.timings_ms()
        mine = torch.tensor([[float(np.mean([t[k] for t in st])) for k in range(3)]], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        extras["sharded_rccl"] = {
            "host": "pg_embedding_amd/sharded.py: one process per GPU, one packed all-gather over RCCL + merge kernel",
            "rows_per_shard": rows_per, "queries_per_step": 1024, "steps": 5, "queries_per_s": 1024 * 5 / float(el.item()),
            "ms_per_step": float(el.item()) / 5 * 1e3, "exchange_bytes_per_rank": block_bytes(1024, args.ef),
            "step_breakdown_ms_per_rank": [{"local_search_ms": float(t[0][0]), "exchange_ms": float(t[0][1]), "merge_ms": float(t[0][2])} for t in every],
            "merged_results_sorted_and_full": bool((counts == args.ef).all().item()) and bool((dists[:, 1:] >= dists[:, :-1]).all().item())}
        sh.index.close()
    except Exception as e:                                   # (a rank that fails here leaves the others in a collective: the timer ends them)
        extras["sharded_rccl"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    # ---- (b) row-sharded inside one process on all N devices: rank 0's child, the others wait
    try:
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", "sharded-native", "--gpus", str(world), "--rows", str(250_000 * world),
                   "--nq", "1024", "--steps", "5", "--warmup", "1", "--dim", str(args.dim), "--hnsw-m", str(args.m), "--efc", str(args.efc),
                   "--ef", str(args.ef), "--timeout", "150"]
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                                   "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE")}
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=160)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                extras["sharded_native"] = json.loads(lines[-1]) if lines else {"error": f"rc {r.returncode}: " + (r.stderr or r.stdout)[-400:]}
            except subprocess.TimeoutExpired:
                extras["sharded_native"] = {"error": "child did not finish within 160 s"}
        dist.barrier()
    except Exception as e:
        extras["sharded_native"] = {"error": f"{type(e).__name__}: {e}"[:400]}
    done.set()
    if rank == 0:
        result["multi_gpu_extras"] = extras
        emit(result)


def host_pointer_batch(args, ix, Q, labels0):
    """SURVEY.md 8(d) "report H2D separately": the whole batch through hnsw_gpu_search_batch — host pointers in, host pointers out,
    the shape a C caller without device buffers uses — with the library's own HIP events around its three steps (upload of the
    queries / search kernel / download of labels + distances + counts).  Twice: ordinary (pageable) host arrays, and buffers from
    hnsw_gpu_host_alloc (pinned: the copies are plain DMA).  Never `value`: `value` has its inputs resident in HBM."""
    import ctypes as C
    import numpy as np
    import torch
    from pg_embedding_amd._lib import check
    nq, ef, dim = Q.shape[0], args.ef, args.dim
    Qh = Q.cpu().numpy()
    out = {"queries": nq, "bytes_up": int(nq * dim * 4), "bytes_down": int(nq * ef * 12 + nq * 4)}

    def run(qp, lp, dp, cp, reps=4):
        best = None
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(ix.L.hnsw_gpu_search_batch(ix._h, qp, nq, ef, lp, dp, cp), "hnsw_gpu_search_batch")
            wall = (time.perf_counter() - t0) * 1e3
            up, k, down = ix.last_batch_ms()
            if best is None or wall < best["call_ms"]:
                best = {"call_ms": wall, "ms_h2d": up, "ms_kernel": k, "ms_d2h": down, "queries_per_s": nq / wall * 1e3}
        return best
    lab = np.empty((nq, ef), np.uint64); dist = np.empty((nq, ef), np.float32); cnt = np.empty(nq, np.uint32)
    out["pageable_host_memory"] = run(Qh.ctypes.data, lab.ctypes.data, dist.ctypes.data, cnt.ctypes.data)
    out["results_identical_to_the_device_pointer_form"] = bool((lab.view(np.int64) == labels0.cpu().numpy()).all())
    sizes = [nq * dim * 4, nq * ef * 8, nq * ef * 4, nq * 4]
    ptrs = [ix.L.hnsw_gpu_host_alloc(b) for b in sizes]
    try:
        if all(ptrs):
            C.memmove(ptrs[0], Qh.ctypes.data, sizes[0])
            out["pinned_host_memory"] = run(*ptrs)
    finally:
        for p in ptrs:
            if p:
                ix.L.hnsw_gpu_host_free(p)
    return out


def two_streams(ix, Qs, ef, labels_want, dev, reps=12):
    """Launches of the same batch alternating on two search contexts / two streams: the next launch's walks fill the slots the previous
    one frees while its last walks drain (what a caller that has the next batch ready gets; ONE call cannot do it for itself — splitting
    a batch in the library was measured slower, profiles/r4c_tail_split_rejected.txt).  Wall-clock over `reps` launches."""
    import torch
    import pg_embedding_amd as pg
    ctxs = [pg.SearchContext(ix), pg.SearchContext(ix)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    outs = [ix.search_torch(Qs, ef), ix.search_torch(Qs, ef)]
    torch.cuda.synchronize()
    for i in range(2):
        ctxs[i].search_torch(Qs, ef, outs[i], streams[i])
    torch.cuda.synchronize()
    # three rounds, each reported: the rounds directly behind seconds of sustained load (the index build) run at whatever clocks that load
    # left — the VALU-bound narrow-row kernel feels it, the HBM-bound one does not (profiles/r4x_launches_back_to_back.txt); the median counts
    rounds = []
    for _ in range(3):
        tp = time.perf_counter()
        for i in range(reps):
            ctxs[i & 1].search_torch(Qs, ef, outs[i & 1], streams[i & 1])
        torch.cuda.synchronize()
        rounds.append(int(Qs.shape[0]) * reps / (time.perf_counter() - tp))
    ok = bool((outs[0]["labels"] == labels_want).all().item() and (outs[1]["labels"] == labels_want).all().item())
    for c in ctxs:
        c.close()
    return {"queries_per_launch": int(Qs.shape[0]), "launches": reps, "streams": 2, "queries_per_s": sorted(rounds)[1],
            "queries_per_s_of_each_round": rounds, "results_identical": ok}


def side_cases(dev):
    """(name, dims, m, metric, row generator, queries per launch) of BASELINE.json configs 2, 3 and 5 and the two stress datasets"""
    import torch
    from pg_embedding_amd.datasets import gmm_torch

    def rows_gmm(cnt, dim, stream):
        return gmm_torch(cnt, dim, k=1000, sigma=0.3, seed=42, stream=stream, device=dev)

    def rows_sift(cnt, dim, stream):          # SIFT-1M stand-in: clustered integers in [0, 218] stored as fp32
        return torch.clamp(torch.round(40.0 + 35.0 * rows_gmm(cnt, dim, stream)), 0, 218)

    def rows_iid(cnt, dim, stream):
        g = torch.Generator(device=dev)
        g.manual_seed(4242 + stream)
        return torch.randn((cnt, dim), generator=g, device=dev, dtype=torch.float32)

    def rows_lowrank(cnt, dim, stream):
        g = torch.Generator(device=dev)
        g.manual_seed(777)
        basis = torch.randn((16, dim), generator=g, device=dev, dtype=torch.float32)
        g.manual_seed(778 + stream)
        out = torch.empty((cnt, dim), device=dev, dtype=torch.float32)
        for i in range(0, cnt, 1 << 18):
            m = min(1 << 18, cnt - i)
            out[i:i + m] = torch.randn((m, 16), generator=g, device=dev) @ basis + 0.05 * torch.randn((m, dim), generator=g, device=dev)
        return out

    return [
        ("C2_sift_like_1Mx128_l2_m16", 128, 16, "l2", rows_sift, 40000),
        ("C3_1Mx768_cosine_m32", 768, 32, "cosine", rows_gmm, 40000),
        ("C5_1Mx1536_cosine_m32_Q1024", 1536, 32, "cosine", rows_gmm, 1024),
        ("stress_iid_1Mx768_l2_m16", 768, 16, "l2", rows_iid, 10000),
        ("stress_lowrank16_1Mx768_l2_m16", 768, 16, "l2", rows_lowrank, 40000),
    ]


def build_side_config(args, case, dev, local):
    """index + queries of one side configuration, exactly as side_configs (and --profile-config) search them"""
    import torch
    import pg_embedding_amd as pg
    name, dim, m, metric, gen, nq = case
    n = min(args.n, 1_000_000)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE}[metric]
    X = gen(n, dim, 0)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, m, args.efc, args.ef, func), n, device=local)
    ix.append_torch(X)
    del X
    ix.link(0, n, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    return ix, gen(nq, dim, 1)


def side_configs(args, dev, local):
    """BASELINE.json configs 2, 3 and 5 at 1M rows, and two stress datasets for the headline shape: i.i.d. Gaussian rows
    (no cluster structure: E_q several times larger, recall far below the gate — SURVEY.md §8d says report it, never gate
    on it) and low-rank rows (16-d latent + noise).  One launch size each, kernel time from the library's HIP events."""
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k
    n = min(args.n, 1_000_000)
    # every leg in front of this one ends by handing torch's cached device blocks back; with those legs switched off (--serial-rows 0
    # --hostile-rows 0) it did not happen, and the narrow-row launch was 40 % slower on every launch in two runs of four (profiles/r5af_*:
    # cause open, this call not shown to be the remedy): start from the same allocator state whatever ran before
    if not args.alloc_probe:
        torch.cuda.empty_cache()

    cases = side_cases(dev)
    res = {}
    for name, dim, m, metric, gen, nq in cases:
        t0 = time.time()
        ix, Q = build_side_config(args, (name, dim, m, metric, gen, nq), dev, local)
        t_build = time.time() - t0
        out = ix.search_torch(Q, args.ef, stats=True)
        torch.cuda.synchronize()
        st = out["stats"].cpu().numpy().astype(np.int64)
        cnt = out["counts"].cpu().numpy().astype(np.int64)
        bq = alg_bytes(st, cnt, dim, m)
        nrec = min(256, nq)
        truth, _ = ix.bruteforce_torch(Q[:nrec].contiguous(), 10, mfma=True)
        rec = recall_at_k(out["labels"][:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
        mfma = None
        if name.startswith("C5"):
            # BASELINE config 5, "ef x dims as MFMA GEMM": exhaustive scoring of the whole Q=1024 batch against every row as an
            # f32 MFMA contraction (filter) + canonical re-scoring of the survivors = bit-identical to the canonical scan
            # (csrc/device_bf_mfma.h); flops = 2 * Q * N * D over the GEMM kernel's own HIP-event time
            from pg_embedding_amd._lib import gpu_lib
            gl = gpu_lib()
            best_ms, best_total, best_mhz, all_ms = 1e30, 1e30, 0.0, []
            for _ in range(5):                                # (the first call runs while the shader clock is still ramping up: 2.2-2.3 GHz)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ti, td = ix.bruteforce_torch(Q, 10, mfma=True)
                torch.cuda.synchronize()
                best_total = min(best_total, (time.perf_counter() - t1) * 1e3)
                ms_k = float(gl.hnsw_gpu_last_bruteforce_gemm_ms())
                all_ms.append(ms_k)
                if ms_k < best_ms:
                    best_ms, best_mhz = ms_k, float(gl.hnsw_gpu_last_bruteforce_clock_mhz())
            flops = 2.0 * nq * n * ((dim + 3) // 4 * 4)
            si, sd = ix.bruteforce_torch(Q[:64].contiguous(), 10)          # the canonical scan on a slice
            mfma = {"queries": nq, "block_tile": int(gl.hnsw_gpu_last_bruteforce_tile()), "gemm_kernel_ms": best_ms,
                    "gemm_kernel_ms_all_calls": all_ms, "gemm_kernel_ms_median": float(np.median(all_ms)), "shader_clock_mhz_in_best_call": best_mhz, "tflops": flops / best_ms / 1e9, "peak_f32_mfma_tflops": 157.3,
                    "frac": flops / best_ms / 1e9 / 157.3, "whole_call_ms": best_total,
                    "identical_to_canonical_scan": bool((si == ti[:64]).all().item() and
                                                        (sd.view(torch.int32) == td[:64].view(torch.int32)).all().item())}
        ms = []
        for _ in range(13):                                   # 1 warm-up + 12 timed: a bimodal kernel shows in min / max
            ix.search_torch(Q, args.ef, out=out)
            ms.append(ix.last_search_ms())
        kms = float(np.median(ms[1:]))
        timed_kernel = ix.last_search_kernel()                 # (read BEFORE the traced launch below: that one runs another instantiation)
        clock_mhz = ix.last_search_clock_mhz()                 # shader clock the last timed launch ran at (its first wave's stamps)
        place = ix.placement()
        probe = None
        if args.alloc_probe and name.startswith("C2"):
            # diagnostic (profiles/r5af_*): the cache was NOT emptied in front of this index.  Which allocations carry the slow state —
            # torch's (query and result tensors carved from the blocks it kept) or the library's (index, workspace)?  Time the same
            # index again (a) with query / result tensors re-created after torch gave its blocks back, (b) through a second search
            # context (its own workspace, allocated now)
            Qh, lab0 = Q.cpu(), out["labels"].clone()
            del out, Q
            torch.cuda.empty_cache()
            Q = Qh.to(dev)
            out = ix.search_torch(Q, args.ef, stats=True)
            m2 = []
            for _ in range(9):
                ix.search_torch(Q, args.ef, out=out)
                m2.append(ix.last_search_ms())
            probe = {"slow_state_kernel_ms": kms, "after_empty_cache_and_new_query_and_result_tensors_ms": float(np.median(m2[1:])),
                     "results_identical": bool((out["labels"] == lab0).all().item()), "torch_reserved_GB": torch.cuda.memory_reserved() / 2**30}
            print("alloc probe: " + json.dumps(probe), file=sys.stderr, flush=True)
        ach = float(bq.sum()) / (kms * 1e-3) / 1e9
        tr = finish_trace_roof(trace_roof(ix, Q, args.ef), kms, float(bq.sum()))   # replay of this launch's own row trace
        res[name] = {"rows": n, "dims": dim, "m": m, "metric": metric, "efsearch": args.ef, "queries_per_launch": nq,
                     "queries_per_s": nq / kms * 1e3, "kernel_ms_per_launch": kms,
                     "kernel_ms_min_median_max": [float(min(ms[1:])), kms, float(max(ms[1:]))], "launches_timed": len(ms) - 1,
                     "achieved_GBps": ach,
                     "frac_of_8TBps": ach / HBM_PEAK_GBS, "replay_GBps": tr["replay_GBps"], "frac_of_replay": tr["frac_of_replay"],
                     "reads_beyond_infinity_cache_reach": tr["reads_beyond_infinity_cache_reach"],
                     "hbm_lower_bound_GBps": tr["hbm_lower_bound_GBps"], "evals_per_query": float(st[:, 0].mean()),
                     "hops_per_query": float(st[:, 1].mean()), "recall_at_10": rec, "kernel": timed_kernel,
                     "datagen_plus_build_seconds": t_build,
                     # which state of the device / of the process's allocations the launches ran in (profiles/r5af_*: a narrow-row launch
                     # 40 % slower on every launch of some processes): the clock the kernel measured itself, where the mirror sits
                     "shader_clock_mhz": clock_mhz, "mirror_arrays_on_2MiB_boundaries": place["aligned_2MiB"],
                     "placement": {k: [hex(v[0]), v[1]] for k, v in place.items() if isinstance(v, tuple)}}
        key = name.split("_")[0]
        if key in ("C2", "C3", "C5"):
            traffic, tsrc = pmc_traffic_for(key, timed_kernel, {"n": n, "dim": dim, "m": m, "efc": args.efc, "ef": args.ef, "nq": nq, "metric": metric})
            res[name]["traffic"] = traffic
            res[name]["traffic_source"] = tsrc
            res[name]["traffic_over_algorithmic"] = (traffic / float(bq.sum())) if traffic else None
        if mfma:
            res[name]["exhaustive_mfma_gemm"] = mfma
        if probe:
            res[name]["alloc_probe"] = probe
        if nq >= 10000:
            # the same launches back to back on two streams: what the drain of a launch's last walks costs a caller that has no next batch
            # ready (frac_of_8TBps above) and what one that has gets (here)
            ts = two_streams(ix, Q, args.ef, out["labels"], dev)
            ts["frac_of_8TBps"] = float(bq.sum()) * ts["queries_per_s"] / nq / 1e9 / HBM_PEAK_GBS
            res[name]["two_streams"] = ts
        ix.close()
        del ix, out, Q
        torch.cuda.empty_cache()
    # ---- BASELINE config 4 functionally, on ONE device: 10M x 768 as 8 row shards (independent graphs), the native one-process
    # sharded search of the C ABI (per-shard searchKnn + device merge).  One GPU does the work of eight here; the 8-GPU form is
    # `bench.py --mode sharded --gpus 8` (one packed all-gather over RCCL).
    if args.n >= 1_000_000:
        from pg_embedding_amd.index import LocalShardedIndex
        n4, shards, nq4, dim = 10_000_000, 8, 1024, 768
        t0 = time.time()
        meta = pg.make_meta(dim, 16, args.efc, args.ef, pg.DIST_L2)
        idx = []
        for r in range(shards):
            lo, hi = n4 * r // shards, n4 * (r + 1) // shards
            rows = gmm_torch(hi - lo, dim, k=1000, sigma=0.3, seed=42, stream=100 + r, device=dev)
            ix = pg.GpuIndex.empty(meta, hi - lo, device=local)
            ix.append_torch(rows, torch.arange(lo, hi, dtype=torch.int64, device=dev))
            del rows
            ix.link(0, hi - lo, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
            idx.append(ix)
        torch.cuda.synchronize()
        t_build = time.time() - t0
        sh = LocalShardedIndex(idx)
        Q = gmm_torch(nq4, dim, k=1000, sigma=0.3, seed=42, stream=1, device=dev)
        ml, md, mc = sh.search_torch(Q, args.ef)
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ml, md, mc = sh.search_torch(Q, args.ef)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        dt = float(np.median(ts[1:]))
        cand_i, cand_d = [], []
        for r, ix in enumerate(idx):                      # exhaustive truth per shard (MFMA scorer), merged
            ti, td = ix.bruteforce_torch(Q[:256].contiguous(), 10, mfma=True)
            cand_i.append(ti.long() + n4 * r // shards)
            cand_d.append(td)
        ci, cd = torch.cat(cand_i, 1), torch.cat(cand_d, 1)
        truth = torch.gather(ci, 1, torch.argsort(cd, dim=1)[:, :10])
        rec = recall_at_k(ml[:256].cpu().numpy(), truth.cpu().numpy(), 10)
        res["C4_10Mx768_l2_8shards_on_ONE_device"] = {
            "rows": n4, "dims": dim, "m": 16, "shards": shards, "queries_per_batch": nq4, "efsearch": args.ef,
            "entry_point": "hnsw_gpu_sharded_search_dev (one process, per-shard search + strided device merge)",
            "ms_per_batch_all_shards_plus_merge": dt * 1e3, "queries_per_s": nq4 / dt, "recall_at_10": rec,
            "results_per_query_full": bool((mc == args.ef).all().item()), "datagen_plus_build_seconds": t_build}
        sh.close()
        for ix in idx:
            ix.close()
        del idx, Q
        torch.cuda.empty_cache()
    res["_note"] = ("achieved_GBps = algorithmic bytes (SURVEY 8d) / kernel time, frac_of_8TBps = that over the nominal HBM peak; rows that "
                    "many queries of a launch share are served by L2 / Infinity Cache, so the figure can exceed what HBM delivers "
                    "(the i.i.d. set: every walk crosses the same hub rows) -- replay_GBps is the launch's own row trace gathered again "
                    "with nothing in between (frac_of_replay <= 1 by construction), hbm_lower_bound_GBps the reads no cache can hold "
                    "over the kernel's time; roofline.hbm_only is the cache-hostile table of the headline shape")
    return res


def graph_figures(ix, Q, ef, nrec, truth, dim, m):
    """E_q, H_q, recall@10, mean degree and q/s of one built index (the figures the two builds are compared on)"""
    import numpy as np
    import torch
    from pg_embedding_amd.datasets import recall_at_k
    out = ix.search_torch(Q, ef, stats=True)
    torch.cuda.synchronize()
    st = out["stats"].cpu().numpy().astype(np.int64)
    cnt = out["counts"].cpu().numpy().astype(np.int64)
    rec = recall_at_k(out["labels"][:nrec].cpu().numpy(), truth, 10)
    ms = []
    for _ in range(4):
        ix.search_torch(Q, ef, out=out)
        ms.append(ix.last_search_ms())
    kms = float(np.median(ms[1:]))
    deg = ix.export_flat().reshape(ix.count, -1)[:, :4].copy().view(np.uint32).ravel()      # the count word of every element image
    bq = alg_bytes(st, cnt, dim, m)
    return {"evals_per_query": float(st[:, 0].mean()), "hops_per_query": float(st[:, 1].mean()), "recall_at_10": rec,
            "mean_degree": float(deg.mean()), "full_lists": float((deg == 2 * m).mean()),
            "alg_bytes_per_query": float(bq.mean()), "queries_per_s": Q.shape[0] / kms * 1e3, "kernel_ms_per_launch": kms}


REF_GRAPH = os.path.join(ROOT, "oracle", "_ref", "serial_graph_{n}x{dim}_m{m}_efc{efc}_l2.npy")


def reference_graph_vs_batched(args, dev, local, func, path, nq, serial=None):
    """The headline table as the REFERENCE ITSELF builds it: oracle/_ref's serial hnsw_bind_point over the rows (hnswalg.cpp:279-291), made
    once on a host core by tests/experiments/make_ref_serial_graph.py (17 minutes for 1M x 768; the link words travel in oracle/_ref/ like
    the reference binaries), uploaded byte for byte and searched beside the BATCHED device build of the same rows with the same queries.
    Rows and queries are the numpy generator's (the same bytes on every box).  `serial`: the figures of the reference's graph when the
    headline legs have already measured them (the headline IS that graph whenever it is present): only the batched build is made then."""
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm, gmm_chunks
    n = args.n
    Q = gmm(nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1)
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    Qd = torch.from_numpy(Q).to(dev)
    nrec = min(1000, nq)
    res = {"rows": n, "dims": args.dim, "m": args.m, "efconstruction": args.efc, "efsearch": args.ef, "queries_per_launch": nq,
           "serial_graph_built_by": "oracle/_ref = the unmodified reference (hnsw_bind_point row by row on one host core), " + os.path.relpath(path, ROOT)}
    truth = None
    if serial is None:
        ix, _, fig = build_index_on_reference_graph(args, path, dev, local, func)
        truth = ix.bruteforce_torch(Qd[:nrec].contiguous(), 10, mfma=True)[0].cpu().numpy()
        res["serial"] = graph_figures(ix, Qd, args.ef, nrec, truth, args.dim, args.m)
        ix.close()
    else:                                                       # (the headline legs searched exactly this graph with exactly these queries)
        res["serial"] = dict(serial)
    ix = pg.GpuIndex.empty(meta, n, device=local)
    for _, x in gmm_chunks(n, args.dim, k=args.clusters, sigma=0.3, seed=42, chunk=1 << 16):
        ix.append_torch(torch.from_numpy(x).to(dev))
    torch.cuda.synchronize()
    t0 = time.time()
    ix.link(0, n, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    if truth is None:
        truth = ix.bruteforce_torch(Qd[:nrec].contiguous(), 10, mfma=True)[0].cpu().numpy()
    res["batched"] = graph_figures(ix, Qd, args.ef, nrec, truth, args.dim, args.m)
    res["batched"]["build_seconds"] = t_build
    ix.close()
    del ix, Qd
    torch.cuda.empty_cache()
    return res


def serial_vs_batched(args, dev, local, func, n, nq=10_000, head_serial=None):
    """Is the headline index the reference's workload?  The reference builds its graph by serial hnsw_bind_point calls
    (hnswalg.cpp:279-291); the bench builds with the BATCHED device builder (hnsw_gpu_index_link, batches <= 4096: a different graph
    by construction).  Both graphs of the same rows are searched with the same queries: E_q, H_q, recall@10, mean degree and q/s side
    by side.  Where the reference's own serial graph of the headline table travels with the tree (reference_graph_vs_batched) that one
    is used, at full size; otherwise `n` rows are built twice on the device — serial (max_batch = 1: graph bytes == the oracle's serial
    build, tests/test_gpu_build.py and tests/test_gpu_insert_fullsize.py) and batched."""
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch
    path = REF_GRAPH.format(n=args.n, dim=args.dim, m=args.m, efc=args.efc)
    if func == pg.DIST_L2 and os.path.exists(path):
        res = reference_graph_vs_batched(args, dev, local, func, path, args.nq, head_serial)
    else:
        clusters = max(10, args.clusters * n // max(args.n, 1)) if n < args.n else args.clusters
        X = gmm_torch(n, args.dim, k=clusters, sigma=0.3, seed=42, device=dev)
        Q = gmm_torch(nq, args.dim, k=clusters, sigma=0.3, seed=42, stream=1, device=dev)
        meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
        nrec = min(1000, nq)
        res = {"rows": n, "dims": args.dim, "m": args.m, "efconstruction": args.efc, "efsearch": args.ef, "clusters": clusters,
               "queries_per_launch": nq, "serial_graph_built_by": "the device, hnsw_gpu_index_link(max_batch = 1): the oracle's bytes"}
        truth = None
        for name, mb in (("serial", 1), ("batched", args.max_batch)):
            ix = pg.GpuIndex.empty(meta, n, device=local)
            ix.append_torch(X)
            torch.cuda.synchronize()
            t0 = time.time()
            ix.link(0, n, mb, args.ratio if mb != 1 else 0, torch.cuda.current_stream(dev).cuda_stream)
            torch.cuda.synchronize()
            t_build = time.time() - t0
            if truth is None:
                truth = ix.bruteforce_torch(Q[:nrec].contiguous(), 10, mfma=True)[0].cpu().numpy()
            res[name] = graph_figures(ix, Q, args.ef, nrec, truth, args.dim, args.m)
            res[name]["build_seconds"] = t_build
            res[name]["max_batch"] = mb
            ix.close()
            del ix
        del X, Q
        torch.cuda.empty_cache()
    a, b = res["serial"], res["batched"]
    rel = lambda k: (b[k] - a[k]) / a[k]
    res["batched_minus_serial_relative"] = {k: rel(k) for k in ("evals_per_query", "hops_per_query", "recall_at_10", "mean_degree",
                                                               "alg_bytes_per_query", "queries_per_s")}
    res["within_2_percent"] = bool(abs(rel("evals_per_query")) <= 0.02 and abs(rel("recall_at_10")) <= 0.02)
    return res


def serial_insert(args, dev):
    """The other side of the boundary (SURVEY.md §8 'next' row: hnsw_bind_point): one row at a time into a 20 000-row mirror of the
    headline shape — hnsw_gpu_index_insert_one (append + the insert's own walk + link + the changed lists back, one call) and
    hnsw_gpu_index_insert_candidates behind a traced walk (what the validated cache of the unmodified glue runs per insert).
    Wall-clock medians of the calls on this host core; not part of `value`."""
    import ctypes as C
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd._lib import check
    from pg_embedding_amd.datasets import gmm_torch
    n, extra = 20000, 400
    X = gmm_torch(n + extra, args.dim, k=100, sigma=0.3, seed=7, stream=0, device=dev).cpu().numpy()
    meta = pg.make_meta(args.dim, args.m, 64, 64, pg.DIST_L2)
    ix = pg.GpuIndex.empty(meta, n + extra)
    ix.append(X[:n])
    ix.link(0, n)
    torch.cuda.synchronize()
    maxM = int(meta.maxM)
    mine = (C.c_uint32 * (maxM + 1))()
    others = (C.c_uint32 * (maxM * (maxM + 1)))()
    t_one, t_cand, t_walk = [], [], []
    for i in range(extra):
        p = np.ascontiguousarray(X[n + i])
        if i < extra // 2:
            t0 = time.perf_counter()
            check(ix.L.hnsw_gpu_index_insert_one(ix._h, p.ctypes.data, n + i, n + i, mine, others), "insert_one")
            t_one.append(time.perf_counter() - t0)
        else:
            t0 = time.perf_counter()
            ci, cd, pops, nev = ix.search_trace(p, 64, base=True)
            t1 = time.perf_counter()
            ci32 = np.ascontiguousarray(ci.astype(np.uint32)); cd32 = np.ascontiguousarray(cd, dtype=np.float32)
            t2 = time.perf_counter()
            check(ix.L.hnsw_gpu_index_insert_candidates(ix._h, p.ctypes.data, n + i, n + i, ci32.ctypes.data, cd32.ctypes.data, len(ci32), mine, others),
                  "insert_candidates")
            t_cand.append(time.perf_counter() - t2); t_walk.append(t1 - t0)
    paths = (C.c_uint64 * 2)()
    ix.L.hnsw_gpu_insert_path_counts(paths)
    ix.close()
    ms = lambda ts: float(np.median(ts)) * 1e3
    res = {"mirror": f"{n}x{args.dim} l2 m={args.m} efconstruction=64 (device build)", "rows_inserted": extra,
           "insert_one_ms_median": ms(t_one), "insert_candidates_ms_median": ms(t_cand), "traced_walk_ef64_ms_median": ms(t_walk),
           "two_launch_inserts": int(paths[0]), "general_path_inserts": int(paths[1])}
    # ---- the fallback of the single insert: more candidates than the two-launch form keeps in one wavefront's registers
    # (max(efconstruction, maxM + 1) > 512, csrc/device_insert.h INS_MAX_SIDE) go through the general builder path
    n2, extra2, m2, efc2 = 8000, 60, 40, 600
    meta2 = pg.make_meta(args.dim, m2, efc2, 64, pg.DIST_L2)
    ix2 = pg.GpuIndex.empty(meta2, n2 + extra2)
    ix2.append(X[:n2])
    ix2.link(0, n2)
    torch.cuda.synchronize()
    maxM2 = int(meta2.maxM)
    mine2 = (C.c_uint32 * (maxM2 + 1))()
    others2 = (C.c_uint32 * (maxM2 * (maxM2 + 1)))()
    before = (C.c_uint64 * 2)()
    ix2.L.hnsw_gpu_insert_path_counts(before)
    t_gen2 = []
    for i in range(extra2):
        p = np.ascontiguousarray(X[n + i])
        t0 = time.perf_counter()
        check(ix2.L.hnsw_gpu_index_insert_one(ix2._h, p.ctypes.data, n2 + i, n2 + i, mine2, others2), "insert_one (general path)")
        t_gen2.append(time.perf_counter() - t0)
    after = (C.c_uint64 * 2)()
    ix2.L.hnsw_gpu_insert_path_counts(after)
    ix2.close()
    res["general_path"] = {"mirror": f"{n2}x{args.dim} l2 m={m2} efconstruction={efc2} (device build)", "rows_inserted": extra2,
                           "insert_one_ms_median": ms(t_gen2), "two_launch_inserts": int(after[0] - before[0]),
                           "general_path_inserts": int(after[1] - before[1])}
    return res


def hostile_setup(args, dev, local, func):
    """index + queries of the cache-hostile table (see hostile()): returns (index, queries, args with the m / efsearch the recall gate
    needed, what was tried, clusters, datagen seconds, build seconds)"""
    import torch
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k
    import copy
    n = args.hostile_rows
    clusters = max(args.nq, n // 200)
    # The metric's gate (recall@10 >= 0.95) must hold on THIS table too, or the figure is about an easier walk: at m = 16 the
    # one-query-per-cluster launch reaches 0.924, so the table is built with m = 32 ("raise m before touching ef")
    # and, should that not be enough, searched with a wider beam — whatever it takes is said in `workload`.
    hargs = copy.copy(args)
    hargs.m = max(args.m, args.hostile_m)
    ix, t_gen, t_build = build_index(hargs, n, clusters, dev, local, func)
    Q = gmm_torch(args.nq, args.dim, k=clusters, sigma=0.3, seed=42, stream=1, device=dev, distinct_clusters=True)
    nrec = min(500, args.nq)
    truth, _ = ix.bruteforce_torch(Q[:nrec].contiguous(), 10, mfma=True)
    ef_used, tried = args.ef, []
    for ef_try in [args.ef] + [e for e in (160, 192, 256) if e > args.ef]:
        o = ix.search_torch(Q[:nrec].contiguous(), ef_try)
        torch.cuda.synchronize()
        r = recall_at_k(o["labels"][:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
        tried.append({"efsearch": ef_try, "recall_at_10": r})
        ef_used = ef_try
        if r >= 0.95:
            break
    args = copy.copy(args)
    args.ef, args.m = ef_used, hargs.m
    return ix, Q, args, tried, clusters, truth, t_gen, t_build


def hostile(args, dev, local, func):
    """Same kernel, same row width, on a table built to defeat the caches: `--hostile-rows` rows (default 8 M = 24.6 GB, two
    orders of magnitude above the 256 MB Infinity Cache) in clusters of 200, and every query of the launch from a cluster of its
    OWN (a random permutation of the clusters).  Rows still repeat inside a launch — every walk starts at the same entry point,
    and a walk of ~1 800 rows crosses its neighbours' clusters; a table in which 40 000 walks never meet would need > 70 M rows —
    but almost never within cache reach: the launch's own trace says how many reads had their row read less than 256 MB of
    traffic earlier (`reads_beyond_infinity_cache_reach` is the complement), and what HBM provably delivered."""
    import numpy as np
    import torch
    from pg_embedding_amd.datasets import recall_at_k
    n = args.hostile_rows
    ix, Q, args, tried, clusters, truth, t_gen, t_build = hostile_setup(args, dev, local, func)
    nrec = min(500, args.nq)
    out = ix.search_torch(Q, args.ef, stats=True)
    torch.cuda.synchronize()
    stats = out["stats"].cpu().numpy().astype(np.int64)
    counts = out["counts"].cpu().numpy().astype(np.int64)
    bq = alg_bytes(stats, counts, args.dim, args.m)
    rec = recall_at_k(out["labels"][:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
    g_rand, g_cfg = random_gather(ix)
    tr_raw = trace_roof(ix, Q, args.ef)
    ms = []
    for _ in range(max(3, args.steps)):
        ix.search_torch(Q, args.ef, out=out)
        ms.append(ix.last_search_ms())
    kms = float(np.mean(ms[1:]))
    ach = float(bq.sum()) / (kms * 1e-3) / 1e9
    tr = finish_trace_roof(tr_raw, kms, float(bq.sum()))
    res = {"workload": f"{n}x{args.dim} fp32 GMM({clusters} clusters, sigma 0.3), {args.metric}, m={args.m}, efsearch={args.ef}, "
                       f"{args.nq} queries/launch, one per cluster",
           "kernel": ix.last_search_kernel(),
           # what HBM provably delivered: the reads no cache on the chip could hold, over the kernel's time
           "achieved": tr["hbm_lower_bound_GBps"], "frac": tr["hbm_lower_bound_GBps"] / HBM_PEAK_GBS,
           "replay_roof": tr["replay_GBps"],
           "algorithmic_GBps": ach, "algorithmic_frac": ach / HBM_PEAK_GBS, "frac_of_replay": tr["frac_of_replay"],
           "trace": tr,
           "random_row_gather_GBps": g_rand, "random_row_gather_config": g_cfg,
           "kernel_ms_per_launch": kms, "queries_per_s": args.nq / kms * 1e3,
           "alg_bytes_per_launch": float(bq.sum()), "evals_per_query": float(stats[:, 0].mean()),
           "hops_per_query": float(stats[:, 1].mean()), "recall_at_10": rec, "recall_gate_0.95_holds": bool(rec >= 0.95),
           "m": args.m, "efsearch": args.ef, "recall_by_efsearch_tried": tried,
           "build_seconds": t_build, "datagen_seconds": t_gen, "shader_clock_mhz": ix.last_search_clock_mhz()}
    traffic, tsrc = pmc_traffic_for("hostile", res["kernel"], {"n": n, "dim": args.dim, "m": args.m, "efc": args.efc, "ef": args.ef, "nq": args.nq, "metric": args.metric})
    res["traffic"], res["traffic_source"] = traffic, tsrc
    res["traffic_over_algorithmic"] = (traffic / res["alg_bytes_per_launch"]) if traffic else None
    ix.close()
    return res


# ------------------------------------------------------------------------------------------ sharded
def main_sharded(args):
    """Row-sharded index (SURVEY.md §8e mode 2, BASELINE config C4): contiguous row ranges, one
    graph per shard, every rank searches the same query batch on its shard, ONE exchange
    (a packed all-gather of the (dist,label) lists over RCCL) and the device merge kernel."""
    import numpy as np
    import torch
    import torch.distributed as dist

    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k
    from pg_embedding_amd.sharded import ShardedIndex, block_bytes, shard_range

    world, rank, local, use_dist, backend = init_ranks(args)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]
    lo, hi = shard_range(args.n, world, rank)
    clusters = max(args.clusters, args.n // 1000)
    t0 = time.time()
    # every rank generates ITS rows of one global mixture: same centres everywhere (seed), own stream
    rows = gmm_torch(hi - lo, args.dim, k=clusters, sigma=0.3, seed=42, stream=100 + rank, device=dev)
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    sh = ShardedIndex.build(rows, lo, meta, device=local, max_batch=args.max_batch, ratio=args.ratio)
    torch.cuda.synchronize()
    t_build = time.time() - t0
    nq = args.nq
    Q = gmm_torch(nq, args.dim, k=clusters, sigma=0.3, seed=42, stream=1, device=dev)   # same on every rank

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # recall@10 of the merged answer against exhaustive search over ALL shards (exact per shard + same merge)
    nrec = min(200, nq)
    tl, td = sh.index.bruteforce_torch(Q[:nrec].contiguous(), 10, mfma=True)
    tlab = (tl.to(torch.int64) + lo)
    tsh = ShardedIndex(index=sh.index, local_search=lambda q, ef: (tlab, td))
    truth, _, _ = tsh.search(Q[:nrec].contiguous(), 10)
    del rows

    # this rank's share of the algorithmic bytes (its own E_q / H_q on its own shard) for its roofline fraction
    st_out = sh.index.search_torch(Q, args.ef, stats=True)
    torch.cuda.synchronize()
    st = st_out["stats"].cpu().numpy().astype(np.int64)
    my_bytes = float(alg_bytes(st, st_out["counts"].cpu().numpy().astype(np.int64), args.dim, args.m).sum())
    del st_out
    merged = (torch.empty((nq, args.ef), dtype=torch.int64, device=dev), torch.empty((nq, args.ef), dtype=torch.float32, device=dev),
              torch.empty(nq, dtype=torch.int32, device=dev))            # the merge's outputs, allocated once
    for _ in range(args.warmup):
        sh.search(Q, args.ef, out=merged)
    barrier()
    sh.record_timing = True
    ex0 = sh.exchanges
    t0 = time.perf_counter()
    for _ in range(args.steps):
        labels, dists, counts = sh.search(Q, args.ef, out=merged)
    barrier()
    elapsed = time.perf_counter() - t0
    exchanges = sh.exchanges - ex0
    local_ms = sh.index.last_search_ms()
    steps_ms = sh.timings_ms()                       # per step on THIS rank: local search / pack + all-gather / merge
    kern_ms = float(np.mean([sh.index.last_search_ms(back) for back in range(min(args.steps, 64))]))     # the search kernel alone, its own HIP events
    mine = torch.tensor([[float(np.mean([t[k] for t in steps_ms])) for k in range(3)] + [kern_ms, my_bytes, float(st[:, 0].mean()), float(st[:, 1].mean())]],
                        dtype=torch.float64, device=dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    if use_dist:
        dist.all_gather(every, mine)
    else:
        every = [mine]
    per_rank_steps = [[float(x) for x in t.flatten().tolist()] for t in every]
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    ok = bool((counts == args.ef).all().item()) and bool((dists[:, 1:] >= dists[:, :-1]).all().item())
    rec = recall_at_k(labels[:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
    if rank == 0:
        print(json.dumps({
            "metric": "queries/sec, row-sharded index, per-shard searchKnn + RCCL top-k merge",
            "value": nq * args.steps / elapsed, "unit": "queries/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"HNSW search, index of {args.n}x{args.dim} rows sharded over {world} GPU(s) "
                                   f"({hi - lo} rows/shard), {args.metric}, m={args.m}, efsearch={args.ef}, "
                                   f"{nq} queries/step (every GPU searches all of them)",
                       "parallelism": f"row-sharded x{world}, one packed all-gather + merge"},
            "ranks": {"world_size_reported_by_backend": world, "backend": backend or "none (single process)",
                      "self_launched": os.environ.get("PGEMB_BENCH_SELF_LAUNCHED") == "1"},
            "exchange": {"collectives_per_step": exchanges / max(args.steps, 1), "bytes_per_rank": block_bytes(nq, args.ef)},
            "local_search_kernel_ms": local_ms,
            # mean over the timed steps, device events on each rank's search stream: where a step's time goes, rank by rank
            # (round 6: the local search writes straight into the rank's packed block — "exchange" is the all-gather alone, no pack kernels)
            "step_breakdown_ms_per_rank": [{"local_search_ms": r[0], "exchange_ms": r[1], "merge_ms": r[2], "search_kernel_ms": r[3],
                                            "alg_bytes_per_launch": r[4], "evals_per_query": r[5], "hops_per_query": r[6],
                                            "roofline_frac": r[4] / (r[3] * 1e-3) / 1e9 / HBM_PEAK_GBS} for r in per_rank_steps],
            "step_breakdown_ms_rank0_per_step": [{"local_search_ms": a, "exchange_ms": b, "merge_ms": c} for a, b, c in steps_ms],
            "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                         "achieved_sum_over_ranks": sum(r[4] / (r[3] * 1e-3) / 1e9 for r in per_rank_steps),
                         "frac_mean_over_ranks": float(np.mean([r[4] / (r[3] * 1e-3) / 1e9 / HBM_PEAK_GBS for r in per_rank_steps])),
                         "note": "every rank's own algorithmic bytes (its E_q / H_q on its shard) over its own search kernel's HIP-event time"},
            "recall_at_10": rec,
            "build_seconds": t_build, "merged_results_sorted_and_full": ok}))
    if use_dist:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------ sharded, one process
def main_sharded_native(args):
    """Row-sharded index inside ONE process (include/hnsw_gpu.h, hnsw_gpu_sharded_*): shard i on device i % --gpus, every shard
    searches every query on its own device and stream, the result lists go straight into the merging device's buffer (peer stores
    over xGMI; a staged peer copy where peer access is refused), one merge kernel.  What a C host — or hnsw_gpu_server — uses for an
    index larger than one GPU; no torch.distributed, no RCCL.  Prints one line with the per-shard search / peer / merge times."""
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm_torch, recall_at_k
    from pg_embedding_amd.index import LocalShardedIndex
    ndev = torch.cuda.device_count()
    ngpu = max(1, min(args.gpus, ndev))
    shards = args.shards if args.shards > 0 else max(ngpu, 1)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]
    meta = pg.make_meta(args.dim, args.m, args.efc, args.ef, func)
    clusters = max(args.clusters, args.n // 1000)
    t0 = time.time()
    idx = []
    for r in range(shards):
        d = r % ngpu
        lo, hi = args.n * r // shards, args.n * (r + 1) // shards
        dev = torch.device("cuda", d)
        with torch.cuda.device(d):
            rows = gmm_torch(hi - lo, args.dim, k=clusters, sigma=0.3, seed=42, stream=100 + r, device=dev)
            ix = pg.GpuIndex.empty(meta, hi - lo, device=d)
            ix.append_torch(rows, torch.arange(lo, hi, dtype=torch.int64, device=dev))
            del rows
            ix.link(0, hi - lo, args.max_batch, args.ratio, torch.cuda.current_stream(dev).cuda_stream)
            idx.append(ix)
    for d in range(ngpu):
        torch.cuda.synchronize(d)
    t_build = time.time() - t0
    home = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    sh = LocalShardedIndex(idx)
    Q = gmm_torch(args.nq, args.dim, k=clusters, sigma=0.3, seed=42, stream=1, device=home)
    for _ in range(max(1, args.warmup)):
        ml, md, mc = sh.search_torch(Q, args.ef)
    torch.cuda.synchronize()
    steps = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ml, md, mc = sh.search_torch(Q, args.ef)
        steps.append(sh.last_ms())                           # (waits for the step: the per-shard events are read between steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # recall@10 of the merged answer against the exhaustive answer over all shards
    nrec = min(200, args.nq)
    ci, cd = [], []
    for r, ix in enumerate(idx):
        d = r % ngpu
        with torch.cuda.device(d):
            ti, td = ix.bruteforce_torch(Q[:nrec].to(torch.device("cuda", d)).contiguous(), 10, mfma=True)
            ci.append((ti.long() + args.n * r // shards).to(home)); cd.append(td.to(home))
    ci, cd = torch.cat(ci, 1), torch.cat(cd, 1)
    truth = torch.gather(ci, 1, torch.argsort(cd, dim=1)[:, :10])
    rec = recall_at_k(ml[:nrec].cpu().numpy(), truth.cpu().numpy(), 10)
    ok = bool((mc == args.ef).all().item()) and bool((md[:, 1:] >= md[:, :-1]).all().item())
    mean = lambda k: [float(np.mean([st[k][i] for st in steps])) for i in range(shards)]
    line = {
        "metric": "queries/sec, row-sharded index in ONE process: per-shard searchKnn on its own device + peer stores + one merge kernel",
        "value": args.nq * args.steps / elapsed, "unit": "queries/s", "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"HNSW search, index of {args.n}x{args.dim} rows in {shards} shard(s) on {ngpu} device(s) of one process, "
                               f"{args.metric}, m={args.m}, efsearch={args.ef}, {args.nq} queries/step (every shard searches all of them)",
                   "parallelism": f"row-sharded x{shards} on {ngpu} device(s), hnsw_gpu_sharded_search_dev (no collective)"},
        "devices_visible": ndev, "shard_devices": steps[-1]["devices"], "direct_peer_stores": steps[-1]["direct_peer_stores"],
        "per_shard_search_ms": mean("search_ms"), "per_shard_peer_ms": mean("peer_ms"),
        "merge_ms": float(np.mean([st["merge_ms"] for st in steps])),
        "recall_at_10": rec, "merged_results_sorted_and_full": ok, "build_seconds": t_build}
    print(json.dumps(line))
    sh.close()
    for ix in idx:
        ix.close()


def kernel_source_digest():
    """sha256 of the sources the search kernels are made of: a traffic figure belongs to exactly one version of them"""
    import hashlib
    h = hashlib.sha256()
    for f in ("device_dist.h", "device_search.h"):
        with open(os.path.join(ROOT, "pg_embedding_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic_for(key, kernel_name, workload):
    """HBM bytes per launch of configuration `key`'s search kernel from rocprofv3 PMC passes of `bench.py --profile-config key`
    ((2*FETCH_SIZE + WRITE_SIZE)*1024: the gfx950 FETCH_SIZE correction of the micro-arch guide).  The counters cannot be
    collected from inside the timed process, so what scripts/profile_configs.sh measured is committed as profiles/traffic.json —
    one entry per configuration — and reported here, with its source and the kernel time of the profiled run, only when it was
    taken for the same workload, the same kernel symbol AND the same bytes of the kernel sources; otherwise null (a stale entry
    is refused, not reported)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            t = json.load(f).get(key)
        if not t:
            return None, f"profiles/traffic.json has no entry for {key}"
        diff = [k for k, v in workload.items() if t["workload"].get(k) != v]
        if diff:
            return None, f"profiles/traffic.json[{key}] is for another workload ({', '.join(diff)} differ): not reported"
        if t.get("kernel") != kernel_name:
            return None, f"profiles/traffic.json[{key}] was taken for kernel {t.get('kernel')}, this run's is {kernel_name}: not reported"
        if t.get("kernel_source_digest") != kernel_source_digest():
            return None, (f"profiles/traffic.json[{key}] was taken for another version of the kernel sources ({t.get('kernel_source_digest')} vs "
                          f"{kernel_source_digest()}): not reported; regenerate with scripts/profile_configs.sh")
        src = f"profiles/traffic.json[{key}] ({t.get('run', 'scripts/profile_configs.sh')}): " + t["source"]
        if t.get("kernel_ms_per_launch_of_that_run"):
            src += f"; kernel time of that run {t['kernel_ms_per_launch_of_that_run']:.3f} ms"
        return float(t["hbm_bytes_per_launch"]), src
    except Exception:
        return None, None


def pmc_traffic(args, kernel_name, graph_built_by):
    """the headline workload's entry ("M"): also tied to which graph was searched (the reference's own or the batched device build)"""
    return pmc_traffic_for("M", kernel_name, {"n": args.n, "dim": args.dim, "m": args.m, "efc": args.efc, "ef": args.ef, "nq": args.nq,
                                              "metric": args.metric, "graph_built_by": graph_built_by})


PROFILE_CONFIGS = ("M", "C2", "C3", "C5", "hostile")


def profile_config_main(args):
    """`bench.py --profile-config M|C2|C3|C5|hostile`: ONE configuration exactly as the bench builds and searches it — its index, its
    queries, `--steps` launches of the timed kernel and nothing else — for the rocprofv3 passes of scripts/profile_configs.sh (the LAST
    `--steps` dispatches of the search kernel are the timed shape).  Prints one JSON line that scripts/make_traffic_json.py keys the
    counters by."""
    import numpy as np
    import torch
    import pg_embedding_amd as pg
    name = args.profile_config
    local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    func = {"l2": pg.DIST_L2, "cosine": pg.DIST_COSINE, "manhattan": pg.DIST_MANHATTAN}[args.metric]
    extra = {}
    if name == "M":
        ref_path = reference_graph_path(args, func)
        if ref_path:
            from pg_embedding_amd.datasets import gmm
            ix, _, _ = build_index_on_reference_graph(args, ref_path, dev, local, func)
            Q = torch.from_numpy(gmm(args.nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1)).to(dev)
        else:
            from pg_embedding_amd.datasets import gmm_torch
            ix, _, _ = build_index(args, args.n, args.clusters, dev, local, func)
            Q = gmm_torch(args.nq, args.dim, k=args.clusters, sigma=0.3, seed=42, stream=1, device=dev)
        wl = {"n": args.n, "dim": args.dim, "m": args.m, "efc": args.efc, "ef": args.ef, "nq": args.nq, "metric": args.metric,
              "graph_built_by": "reference" if ref_path else "device (batched)"}
        ef, m, dim = args.ef, args.m, args.dim
    elif name == "hostile":
        ix, Q, hargs, tried, clusters, truth, _, _ = hostile_setup(args, dev, local, func)
        ef, m, dim = hargs.ef, hargs.m, args.dim
        wl = {"n": args.hostile_rows, "dim": dim, "m": m, "efc": args.efc, "ef": ef, "nq": args.nq, "metric": args.metric}
    else:
        case = [c for c in side_cases(dev) if c[0].split("_")[0] == name][0]
        _, dim, m, metric, _, nq = case
        ix, Q = build_side_config(args, case, dev, local)
        ef = args.ef
        wl = {"n": min(args.n, 1_000_000), "dim": dim, "m": m, "efc": args.efc, "ef": ef, "nq": nq, "metric": metric}
        extra["case"] = case[0]
    out = ix.search_torch(Q, ef, stats=True)
    torch.cuda.synchronize()
    st = out["stats"].cpu().numpy().astype(np.int64)
    cnt = out["counts"].cpu().numpy().astype(np.int64)
    byt = float(alg_bytes(st, cnt, dim, m).sum())
    ms = []
    for _ in range(max(1, args.steps)):
        ix.search_torch(Q, ef, out=out)
        ms.append(ix.last_search_ms())
    kms = float(np.mean(ms))
    print(json.dumps(dict(extra, profile_config=name, kernel=ix.last_search_kernel(), kernel_source_digest=kernel_source_digest(),
                          alg_bytes_per_launch=byt, kernel_ms_per_launch=kms, kernel_ms_all=ms, achieved_GBps=byt / kms / 1e6,
                          frac_of_8TBps=byt / kms / 1e6 / HBM_PEAK_GBS, shader_clock_mhz=ix.last_search_clock_mhz(),
                          evals_per_query=float(st[:, 0].mean()), hops_per_query=float(st[:, 1].mean()), launches=len(ms), workload=wl)), flush=True)
    ix.close()


def cpu_quota():
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def cpu_baseline(args, ix, Q, gpu_labels, gpu_dists, func):
    """oracle/_ref (the unmodified reference distfunc.c + hnswalg.cpp) or, where that was
    not shipped, the C restatement, timed on this box's host cores over a bounded sample of
    the same queries on the identical graph bytes — and the id parity of the device against the reference
    itself on that sample, every mismatch classified (oracle/hnsw_port.c, PortStats.div_*).
    Checker/baseline only."""
    import numpy as np
    import oracle
    import pg_embedding_amd as pg

    raw = ix.export_flat()
    kind = "reference" if oracle.have_ref() else "port"
    Cls = oracle.RefIndex if kind == "reference" else oracle.PortIndex
    cpu = Cls(args.dim, args.m, args.efc, args.ef, func, capacity=args.n)
    cpu.load_raw(raw, args.n)
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
    quota = cpu_quota()
    # cores = what the figure is worth: the threads that ran, capped by the CPU time the box grants the process tree (round 5: 64 threads
    # on a 16-CPU quota were reported as 64 cores)
    cores = threads if quota is None else max(1, min(threads, int(round(quota))))
    res = {
        "value": qpst, "unit": "queries/s", "cores": cores, "threads": threads, "kind": kind,
        "sample": f"{nt} of the {args.nq} queries on {threads} host threads (one query per thread, "
                  f"shared read-only index{'' if cores == threads else f'; the box grants {quota:g} CPUs of time'}); single thread: {n1} queries",
        "single_thread_qps": qps1,
        "eight_thread_qps": qps8,
        "host_cpus": ncores,
        # CPU time the box actually grants this process tree (cgroup v2 cpu.max = quota / period; null = unlimited or unknown): the MI355X
        # boxes of round 4 showed 256 CPUs and granted 16 — a 64-thread figure measured there is 16 CPUs' worth
        "host_cpu_quota_cpus": quota,
    }
    glab = gpu_labels[:nt].cpu().numpy().view(np.uint64)
    same = (rt["labels"] == glab).all(axis=1)
    res["fraction_of_queries_with_identical_ids"] = float(same.mean())
    if kind == "reference":
        # device == canonical-order oracle bit for bit; the oracle shadows every decision of its walk with the
        # reference's own hnsw_dist_func: a query with no diverging decision provably has the reference's ids,
        # a mismatching query has one, and `gap` is how close the two compared distances were (relative)
        del cpu
        port = oracle.PortIndex(args.dim, args.m, args.efc, args.ef, func, capacity=args.n)
        port.load_raw(raw, args.n)
        port.shadow_reference_distances(True)
        p = port.search_many(Qh[:nt], args.ef, nthreads=threads)
        dk, dm = p["div_kind"], p["div_margin"]
        gd = gpu_dists[:nt].cpu().numpy()
        unexplained = (~same) & (dk == 0)
        # ... and directly: the same launch in the debug arithmetic that reproduces the summation order of THIS oracle/_ref build
        # (HNSW_GPU_REF_ORDER=1, csrc/device_dist.h score_rows_ref: L2 with dims % 16 == 0, cosine / Manhattan with dims % 4 == 0) must
        # return the compiled reference's id list for EVERY query of the sample — no classification needed
        ordered = None
        if (func == 0 and args.dim % 16 == 0) or (func in (1, 2) and args.dim % 4 == 0):
            import torch
            pg.config_set("HNSW_GPU_REF_ORDER", 1)          # (the library reads its environment once: a knob is changed by saying so)
            try:
                if args.ef <= 128:
                    oo = ix.search_torch(Q[:nt].contiguous(), args.ef)
                    torch.cuda.synchronize()
                    olab = oo["labels"].cpu().numpy().view(np.uint64)
                    osame = (rt["labels"] == olab).all(axis=1)
                    # does THIS host's oracle/_ref build sum in the order the debug arithmetic restates (gcc 11.4 -Ofast)?  The distance
                    # bits of the first query's results say: equal = the direct comparison is available here; different = another
                    # compiler built _ref, the comparison means nothing and says so instead of disappearing
                    od0 = oo["dists"][0].cpu().numpy()
                    c0 = int(oo["counts"][0].item())
                    meta = ix.meta
                    img = np.frombuffer(memoryview(raw), dtype=np.uint8).reshape(args.n, int(meta.size_data_per_element))
                    rows0 = np.ascontiguousarray(img[olab[0][:c0].astype(np.int64), int(meta.offset_data):int(meta.offset_label)]).view(np.float32)
                    avail = bool((od0[:c0].view(np.uint32) == oracle.ref_dist_many(func, Qh[0], rows0).view(np.uint32)).all())
                    # ... and what that arithmetic costs: the whole batch, timed like the headline (round 6: the reference order runs with
                    # the canonical code's load shape — same bytes, same bytes in flight, a transposed accumulation through LDS on top)
                    of = ix.search_torch(Q, args.ef, stats=True)
                    torch.cuda.synchronize()
                    ost = of["stats"].cpu().numpy().astype(np.int64)
                    obytes = float(alg_bytes(ost, of["counts"].cpu().numpy().astype(np.int64), args.dim, args.m).sum())
                    oms = []
                    for _ in range(4):
                        ix.search_torch(Q, args.ef, out=of)
                        oms.append(ix.last_search_ms())
                    okms = float(np.median(oms[1:]))
                    ordered = {"available": avail, "queries": nt, "queries_with_the_references_id_list": int(osame.sum()),
                               "identical": bool(osame.all()), "kernel": ix.last_search_kernel(),
                               "queries_per_launch": int(Q.shape[0]), "kernel_ms_per_launch": okms, "queries_per_s": Q.shape[0] / okms * 1e3,
                               "achieved_GBps": obytes / okms / 1e6, "frac": obytes / okms / 1e6 / HBM_PEAK_GBS,
                               "evals_per_query": float(ost[:, 0].mean()),
                               "note": "opt-in arithmetic in the reference build's own summation order (one compiler's output: `available` says "
                                       "whether this host's oracle/_ref is that build), canonical load shape + transposed accumulation; `value` "
                                       "stays on the canonical order"}
            finally:
                pg.config_set("HNSW_GPU_REF_ORDER", None)
        if ordered is None:
            ordered = {"available": False, "note": "not applicable to this shape (L2 needs dims % 16 == 0, cosine / Manhattan dims % 4 == 0, ef <= 128)"}
        res["reference_order_mode"] = ordered
        res["parity_vs_reference"] = {
            "queries": nt,
            "device_equals_oracle_bit_exact": bool((p["labels"] == glab).all() and
                                                   (p["dists"].view(np.uint32) == gd.view(np.uint32)).all()),
            "mismatch_count": int((~same).sum()),
            "mismatch_explained_by_near_tie": int(((~same) & (dk != 0)).sum()),
            "mismatch_unexplained": int(unexplained.sum()),
            "largest_unexplained_margin": float(p["margins"][unexplained].max()) if unexplained.any() else None,
            "largest_gap_at_a_mismatching_decision": float(dm[~same].max()) if (~same).any() else 0.0,
            "tolerance": REL_TOL,
            "queries_with_a_diverging_decision": int((dk != 0).sum()),
            "queries_without_one_all_identical": bool(same[dk == 0].all()),
        }
        # the same, as plain scalars of cpu_baseline itself: a record that keeps only scalars (the driver's) must still show that the
        # id lists that differ from the compiled reference's are ALL classified near-ties and that the device equals the oracle
        pv = res["parity_vs_reference"]
        res["parity_queries"] = nt
        res["device_equals_oracle_bit_exact"] = pv["device_equals_oracle_bit_exact"]
        res["mismatch_count"] = pv["mismatch_count"]
        res["mismatch_explained_by_near_tie"] = pv["mismatch_explained_by_near_tie"]
        res["mismatch_unexplained"] = pv["mismatch_unexplained"]
        res["largest_gap_at_a_mismatching_decision"] = pv["largest_gap_at_a_mismatching_decision"]
        res["parity_tolerance"] = REL_TOL
        res["reference_order_available"] = bool(ordered.get("available"))
        res["reference_order_identical_queries"] = ordered.get("queries_with_the_references_id_list")
        res["reference_order_queries"] = ordered.get("queries")
        res["reference_order_queries_per_s"] = ordered.get("queries_per_s")
        res["reference_order_frac"] = ordered.get("frac")
    return res


if __name__ == "__main__":
    main()
