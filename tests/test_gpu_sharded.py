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
