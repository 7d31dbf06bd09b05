"""Row-sharded search on the device: per-shard fused search + merge kernel == oracle per
shard + CPU merge (the parity definition of SURVEY.md §8e)."""
import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from pg_embedding_amd.sharded import ShardedIndex, shard_range

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nshards", [2, 3, 8])
def test_two_shards_on_one_gpu_match_oracle_merge(nshards):
    import torch
    n, dim, m, efc, ef, nq = 6000, 64, 8, 48, 32, 200
    X = gmm(n, dim, k=40, seed=31)
    Q = gmm(nq, dim, k=40, seed=31, stream=1)
    X[n // 2] = X[3]                      # identical rows in different shards: equal distances
    dq = torch.from_numpy(Q).cuda()
    meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
    lists_l, lists_d, per = [], [], []
    for r in range(nshards):
        lo, hi = shard_range(n, nshards, r)
        labels = np.arange(lo, hi, dtype=np.uint64)
        port = oracle.PortIndex(dim, m, efc, ef, pg.DIST_L2)
        port.add(X[lo:hi], labels)
        per.append(port.search_many(Q, ef))
        ix = pg.GpuIndex.from_flat(meta, port.raw(), hi - lo)
        out = ix.search_torch(dq, ef)
        lists_l.append(out["labels"])
        lists_d.append(out["dists"])
        torch.cuda.synchronize()
        ix.close()
    ml, md, mc = pg.merge_topk_torch(torch.stack(lists_l).contiguous(), torch.stack(lists_d).contiguous(), ef)
    ml, md, mc = ml.cpu().numpy().view(np.uint64), md.cpu().numpy(), mc.cpu().numpy()
    for q in range(nq):
        l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
        d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
        order = np.lexsort((l, d))[:ef]
        assert mc[q] == order.size
        assert (ml[q, :order.size] == l[order]).all()
        assert (md[q, :order.size].view(np.uint32) == d[order].view(np.uint32)).all()


@pytest.mark.parametrize("nshards", [1, 2, 5])
def test_native_sharded_search_in_one_process(nshards, gpu_count):
    """hnsw_gpu_sharded_* (C-ABI, no torch.distributed): shards spread round-robin over the visible devices
    (all on device 0 on a 1-GPU box), per-shard search on its own stream writing into the merge device's
    memory, one merge kernel == oracle per shard + CPU merge.  Host-pointer and device-pointer forms."""
    import torch
    n, dim, m, efc, ef, nq = 9000, 96, 8, 48, 40, 300
    X = gmm(n, dim, k=40, seed=41)
    Q = gmm(nq, dim, k=40, seed=41, stream=1)
    X[n - 5] = X[7]                        # identical rows in different shards: equal distances, label order decides
    meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
    shards, per = [], []
    for r in range(nshards):
        lo, hi = shard_range(n, nshards, r)
        port = oracle.PortIndex(dim, m, efc, ef, pg.DIST_L2)
        port.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))
        port.set_deleted(1)                                     # a vacuumed row per shard is filtered before the merge
        per.append(port.search_many(Q, ef))
        shards.append(pg.GpuIndex.from_flat(meta, port.raw(), hi - lo, device=r % max(gpu_count, 1)))
    sh = pg.LocalShardedIndex(shards)
    ml, md, mc = sh.search(Q, ef)
    tl, td, tc = sh.search_torch(torch.from_numpy(Q).cuda(shards[0].device), ef)
    torch.cuda.synchronize()
    assert (tl.cpu().numpy().view(np.uint64) == ml).all() and (tc.cpu().numpy().view(np.uint32) == mc).all()
    assert (td.cpu().numpy().view(np.uint32) == md.view(np.uint32)).all()
    for q in range(nq):
        l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
        d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
        order = np.lexsort((l, d))[:ef]
        assert mc[q] == order.size
        assert (ml[q, :order.size] == l[order]).all()
        assert (md[q, :order.size].view(np.uint32) == d[order].view(np.uint32)).all()
        assert (ml[q, order.size:] == pg.NO_LABEL).all()
    sh.close()
    for s in shards:
        s.close()


def test_native_sharded_build_recall():
    """LocalShardedIndex.build (device insert path per shard): a sharded index is a different graph from the
    monolithic one, so quality is checked by recall@10 against exhaustive search."""
    n, dim, ef = 40000, 64, 64
    X = gmm(n, dim, k=100, seed=9)
    Q = gmm(200, dim, k=100, seed=9, stream=1)
    meta = pg.make_meta(dim, 12, 100, ef, pg.DIST_L2)
    sh = pg.LocalShardedIndex.build(X, meta, 4)
    labels, dists, counts = sh.search(Q, ef)
    assert (counts == ef).all() and (np.diff(dists, axis=1) >= 0).all()
    Qd, Xd = Q.astype(np.float64), X.astype(np.float64)
    d2 = (Qd ** 2).sum(1)[:, None] - 2.0 * Qd @ Xd.T + (Xd ** 2).sum(1)[None, :]
    truth = np.argsort(d2, axis=1)[:, :10]
    from pg_embedding_amd.datasets import recall_at_k
    assert recall_at_k(labels.astype(np.int64), truth, 10) >= 0.95
    shards = sh.shards
    sh.close()
    for s in shards:
        s.close()


def test_sharded_index_world1_builds_and_searches():
    import torch
    n, dim, ef = 20000, 48, 64
    X = torch.from_numpy(gmm(n, dim, k=100, seed=3)).cuda()
    Q = torch.from_numpy(gmm(100, dim, k=100, seed=3, stream=1)).cuda()
    meta = pg.make_meta(dim, 8, 64, ef, pg.DIST_L2)
    sh = ShardedIndex.build(X, 1000, meta)          # labels start at 1000
    labels, dists, counts = sh.search(Q, ef)
    torch.cuda.synchronize()
    assert (counts == ef).all()
    assert int(labels.min()) >= 1000 and int(labels.max()) < 1000 + n
    d = dists.cpu().numpy()
    assert (np.diff(d, axis=1) >= 0).all()


def _two_rank_worker(rank, world, port, out_path):
    import os
    import sys
    import numpy as np
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # two processes share the one GPU of the test box, so the control plane is gloo here; on a
    # real node each rank owns a GPU and the same code runs over RCCL (backend "nccl")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm
    from pg_embedding_amd.sharded import ShardedIndex, shard_range
    n, dim, ef = 24000, 48, 32
    X = gmm(n, dim, k=60, seed=13)
    Q = torch.from_numpy(gmm(300, dim, k=60, seed=13, stream=1)).cuda()
    lo, hi = shard_range(n, world, rank)
    meta = pg.make_meta(dim, 8, 48, ef, pg.DIST_L2)
    sh = ShardedIndex.build(torch.from_numpy(X[lo:hi]).cuda(), lo, meta)
    labels, dists, counts = sh.search(Q, ef)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(out_path, labels=labels.cpu().numpy(), dists=dists.cpu().numpy(), counts=counts.cpu().numpy(),
                 raw0=sh.index.export_flat())
    else:
        np.savez(out_path + ".r1.npz", raw1=sh.index.export_flat())
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_one_collective(tmp_path):
    """world_size=2 for real (two processes, all-gather, device merge): result == oracle per shard
    + CPU merge on the shards' exported graphs."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "merged.npz")
    mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    raws = [got["raw0"], np.load(out + ".r1.npz")["raw1"]]
    n, dim, ef = 24000, 48, 32
    Q = gmm(300, dim, k=60, seed=13, stream=1)
    per = []
    for r in range(2):
        lo, hi = shard_range(n, 2, r)
        p = oracle.PortIndex(dim, 8, 48, ef, pg.DIST_L2)
        p.load_raw(raws[r], hi - lo)
        per.append(p.search_many(Q, ef))
    for q in range(300):
        l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
        d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
        order = np.lexsort((l, d))[:ef]
        assert (got["labels"][q].view(np.uint64) == l[order]).all()
        assert (got["dists"][q].view(np.uint32) == d[order].view(np.uint32)).all()


def test_shards_in_two_processes_meet_in_one_shared_buffer():
    """Row shards that live in DIFFERENT processes (one GPU-owning server per GPU): the merging process allocates the exchange buffer
    (hnsw_gpu_shared_alloc), the other maps it from the 64-byte IPC handle (hnsw_gpu_shared_open) and searches its shard with the
    output pointers inside it; one merge kernel on the owner == oracle per shard + CPU merge.  Both processes share device 0 here
    (the IPC mapping is the same call that crosses xGMI between two GPUs); no torch.distributed, no collective library."""
    import ctypes as C
    import os
    import subprocess
    import sys
    import torch
    from pg_embedding_amd._lib import check, gpu_lib
    n, dim, m, efc, ef, nq, seed, nshards = 8000, 64, 8, 48, 32, 256, 53, 2
    X = gmm(n, dim, k=40, seed=seed)
    Q = gmm(nq, dim, k=40, seed=seed, stream=1)
    L = gpu_lib()
    lab_bytes, dst_bytes, cnt_bytes = nshards * nq * ef * 8, nshards * nq * ef * 4, nshards * nq * 4
    base, handle = C.c_void_p(), C.create_string_buffer(64)
    check(L.hnsw_gpu_shared_alloc(0, lab_bytes + dst_bytes + cnt_bytes, C.byref(base), handle), "hnsw_gpu_shared_alloc")
    peer = subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "shared_gather_peer.py"), handle.raw.hex(), "1",
                             str(nshards), str(n), str(dim), str(m), str(efc), str(ef), str(nq), str(seed)],
                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        per = []
        meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
        for r in range(nshards):
            lo, hi = shard_range(n, nshards, r)
            port = oracle.PortIndex(dim, m, efc, ef, pg.DIST_L2)
            port.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))
            per.append(port.search_many(Q, ef))
            if r == 0:                                           # my own shard: list 0 of the buffer
                ix = pg.GpuIndex.from_flat(meta, port.raw(), hi - lo)
                dq = torch.from_numpy(Q).cuda()
                check(L.hnsw_gpu_search_batch_dev(ix._h, dq.data_ptr(), nq, ef, base.value, base.value + lab_bytes,
                                                  base.value + lab_bytes + dst_bytes, None, None), "hnsw_gpu_search_batch_dev")
                torch.cuda.synchronize()
        line = peer.stdout.readline().strip()                    # the other process's launch has drained
        assert line == "STORED", (line, peer.stderr.read()[-2000:] if peer.poll() is not None else "")
        ol = torch.empty((nq, ef), dtype=torch.int64, device="cuda")
        od = torch.empty((nq, ef), dtype=torch.float32, device="cuda")
        oc = torch.empty((nq,), dtype=torch.int32, device="cuda")
        check(L.hnsw_gpu_merge_topk_dev(0, base.value, base.value + lab_bytes, nshards, nq, ef, ol.data_ptr(), od.data_ptr(), oc.data_ptr(), None),
              "hnsw_gpu_merge_topk_dev")
        torch.cuda.synchronize()
        peer.stdin.write("MERGED\n"); peer.stdin.flush()
        assert peer.stdout.readline().strip() == "CLOSED"
        assert peer.wait(timeout=60) == 0
    finally:
        if peer.poll() is None:
            peer.kill()
    ml, md, mc = ol.cpu().numpy().view(np.uint64), od.cpu().numpy(), oc.cpu().numpy().view(np.uint32)
    for q in range(nq):
        l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
        d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
        order = np.lexsort((l, d))[:ef]
        assert mc[q] == order.size
        assert (ml[q, :order.size] == l[order]).all()
        assert (md[q, :order.size].view(np.uint32) == d[order].view(np.uint32)).all()
    ix.close()
    check(L.hnsw_gpu_shared_free(0, base), "hnsw_gpu_shared_free")
