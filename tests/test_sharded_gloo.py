"""N>1 host logic on CPU: world_size=2 over gloo.  The local search and the merge are injected
(CPU oracle / numpy) — the collective, the layouts and the label globalisation are the product's."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def numpy_merge(labels, dists, ef):
    """reference merge: ef smallest by (dist, label) over all lists, padding dropped"""
    labels = labels.numpy().astype(np.int64).view(np.uint64)
    dists = dists.numpy()
    world, nq, _ = labels.shape
    ol = np.full((nq, ef), np.uint64(0xFFFFFFFFFFFFFFFF))
    od = np.full((nq, ef), np.inf, np.float32)
    oc = np.zeros(nq, np.int32)
    for q in range(nq):
        l = labels[:, q].ravel()
        d = dists[:, q].ravel()
        keep = l != np.uint64(0xFFFFFFFFFFFFFFFF)
        l, d = l[keep], d[keep]
        order = np.lexsort((l, d))[:ef]
        oc[q] = order.size
        ol[q, :order.size] = l[order]
        od[q, :order.size] = d[order]
    return torch.from_numpy(ol.view(np.int64)), torch.from_numpy(od), torch.from_numpy(oc)


def _worker(rank, world, port, n, dim, ef, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from pg_embedding_amd.datasets import gmm
    from pg_embedding_amd.sharded import ShardedIndex, query_slice, shard_range

    X = gmm(n, dim, k=20, seed=5)
    Q = gmm(24, dim, k=20, seed=5, stream=1)
    lo, hi = shard_range(n, world, rank)
    shard = oracle.PortIndex(dim, 6, 32, ef, 0)
    shard.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))        # labels = global row numbers

    def local_search(q, ef_):
        r = shard.search_many(q.numpy(), ef_)
        lab = r["labels"].copy()
        lab[np.arange(ef_)[None, :] >= r["counts"][:, None]] = np.uint64(0xFFFFFFFFFFFFFFFF)
        return torch.from_numpy(lab.view(np.int64)), torch.from_numpy(r["dists"])

    sh = ShardedIndex(local_search=local_search, merge=numpy_merge)
    sh.record_timing = True                       # what bench.py --mode sharded reports per step and per rank
    labels, dists, counts = sh.search(torch.from_numpy(Q), ef)
    tm = sh.timings_ms()
    assert len(tm) == 1 and len(tm[0]) == 3 and all(t >= 0 for t in tm[0]) and sh.exchanges == 1
    # every rank must hold the same merged answer
    ref = [torch.zeros_like(labels) for _ in range(world)]
    dist.all_gather(ref, labels)
    assert all((r == labels).all() for r in ref)
    # replicas + query sharding: slices tile the batch
    qs = [query_slice(24, world, r) for r in range(world)]
    assert qs[0][0] == 0 and qs[-1][1] == 24 and all(qs[i][1] == qs[i + 1][0] for i in range(world - 1))
    if rank == 0:
        np.savez(tmp, labels=labels.numpy(), dists=dists.numpy(), counts=counts.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_search_world2_gloo(tmp_path):
    sys.path.insert(0, ROOT)
    import oracle
    from pg_embedding_amd.datasets import gmm
    from pg_embedding_amd.sharded import shard_range
    n, dim, ef, world = 1200, 24, 16, 2
    out = str(tmp_path / "merged.npz")
    mp.spawn(_worker, args=(world, _free_port(), n, dim, ef, out), nprocs=world, join=True)
    got = np.load(out)
    # expectation: "oracle per shard + CPU merge" (SURVEY.md §8e parity definition)
    X = gmm(n, dim, k=20, seed=5)
    Q = gmm(24, dim, k=20, seed=5, stream=1)
    per = []
    for r in range(world):
        lo, hi = shard_range(n, world, r)
        s = oracle.PortIndex(dim, 6, 32, ef, 0)
        s.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))
        per.append(s.search_many(Q, ef))
    for q in range(24):
        l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
        d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
        order = np.lexsort((l, d))[:ef]
        assert (got["labels"][q].view(np.uint64)[:order.size] == l[order]).all()
        assert (got["dists"][q][:order.size] == d[order]).all()
        assert got["counts"][q] == order.size
        assert (l[order] < n).all()


def test_shard_ranges_are_balanced_and_cover():
    sys.path.insert(0, ROOT)
    from pg_embedding_amd.sharded import shard_range
    for n in (1, 7, 10_000_000):
        for w in (1, 2, 4, 8):
            r = [shard_range(n, w, i) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1
