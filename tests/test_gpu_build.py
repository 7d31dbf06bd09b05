"""Insert path on the device (hnsw_gpu_index_link): serial mode is bit-identical to the
oracle's graph; batched mode is validated by recall and by CPU/GPU agreement on its bytes."""
import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm, recall_at_k
from util import bits

pytestmark = pytest.mark.gpu


def live_image(raw, meta, n):
    """element images with the dead link slots (past `count`) zeroed"""
    img = raw.reshape(n, -1).copy()
    lw = img[:, :meta.offset_data].copy().view(np.uint32)
    for e in range(n):
        lw[e, 1 + lw[e, 0]:] = 0
    img[:, :meta.offset_data] = lw.view(np.uint8)
    return img


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
@pytest.mark.parametrize("dim,m,efc,n", [(24, 4, 16, 1200), (128, 8, 40, 900), (20, 3, 300, 700), (9, 1, 5, 300), (40, 70, 30, 400)])
def test_serial_link_reproduces_the_oracle_graph(func, dim, m, efc, n):
    X = gmm(n, dim, k=30, seed=3 * dim + func)
    labels = (np.arange(n, dtype=np.uint64) * 7 + 5)
    port = oracle.PortIndex(dim, m, efc, 64, func)
    port.add(X, labels)
    meta = pg.make_meta(dim, m, efc, 64, func)
    ix = pg.GpuIndex.empty(meta, n)
    ix.append(X, labels)
    ix.link(0, n, max_batch=1)
    got = ix.export_flat().reshape(n, -1)
    want = live_image(port.raw(), meta, n)
    assert (got == want).all(), f"{(got != want).any(axis=1).sum()} elements differ"
    ix.close()


def test_incremental_serial_link_matches_insert_by_insert():
    """Link in several calls (first > 0), as repeated hnsw_bind_point calls would."""
    dim, m, efc, n = 32, 6, 24, 800
    X = gmm(n, dim, k=20, seed=8)
    port = oracle.PortIndex(dim, m, efc, 64, pg.DIST_L2)
    port.add(X)
    meta = pg.make_meta(dim, m, efc, 64, pg.DIST_L2)
    ix = pg.GpuIndex.empty(meta, n)
    for a, b in [(0, 1), (1, 300), (300, 301), (301, 800)]:
        ix.append(X[a:b])
        ix.link(a, b - a, max_batch=1)
    assert (ix.export_flat().reshape(n, -1) == live_image(port.raw(), meta, n)).all()
    ix.close()


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE])
def test_batched_link_gives_a_searchable_graph(func):
    import torch
    dim, m, efc, n, nq = 96, 12, 64, 30000, 500
    X = gmm(n, dim, k=200, seed=12)
    Q = gmm(nq, dim, k=200, seed=12, stream=1)
    meta = pg.make_meta(dim, m, efc, 64, func)
    ix = pg.GpuIndex.empty(meta, n)
    ix.append(X)
    ix.link(0, n)                                   # default batching
    raw = ix.export_flat()
    cnt = raw.reshape(n, -1)[:, :4].copy().view(np.uint32).ravel()
    assert cnt.max() <= 2 * m and cnt[1:].min() >= 1
    # every link list is duplicate free and in range
    lk = raw.reshape(n, -1)[:, 4:meta.offset_data].copy().view(np.uint32)
    for e in range(0, n, 37):
        l = lk[e, :cnt[e]]
        assert l.max(initial=0) < n and len(set(l.tolist())) == l.size and e not in l
    # recall of the graph vs exhaustive search with the same metric
    dq = torch.from_numpy(Q).cuda()
    truth, tdist = ix.bruteforce_torch(dq, 10)
    labels, dists, counts = ix.search(Q, 128)
    rec = recall_at_k(labels.astype(np.int64), truth.cpu().numpy(), 10)
    assert rec >= 0.95, rec
    # CPU oracle on the exported bytes agrees with the device search bit for bit
    port = oracle.PortIndex(dim, m, efc, 64, func)
    port.load_raw(raw, n)
    want = port.search_many(Q, 128)
    assert (labels == want["labels"]).all() and (bits(dists) == bits(want["dists"])).all()
    ix.close()


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
def test_bruteforce_is_exact(func):
    import torch
    n, dim, nq, k = 5000, 40, 64, 10
    X = gmm(n, dim, k=30, seed=2)
    Q = gmm(nq, dim, k=30, seed=2, stream=1)
    meta = pg.make_meta(dim, 4, 8, 8, func)
    ix = pg.GpuIndex.empty(meta, n)
    ix.append(X)
    idx, dst = ix.bruteforce_torch(torch.from_numpy(Q).cuda(), k)
    idx, dst = idx.cpu().numpy(), dst.cpu().numpy()
    for q in range(nq):
        d = oracle.port_dist_many(func, Q[q], X)
        order = np.lexsort((np.arange(n), d))[:k]          # ties by lower idx
        assert (idx[q] == order).all()
        assert (bits(dst[q]) == bits(d[order])).all()
    ix.close()


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE])
@pytest.mark.parametrize("n,dim,nq,k", [(50000, 200, 300, 10), (20000, 768, 129, 32), (9000, 30, 64, 5),
                                        # rows that end inside a K step (100 floats: the query copy is zero padded to 128, the row's last chunk re-read times zero),
                                        # one query, a table one row past a tile, k = 1; 1536 floats with more queries than one 128-query tile
                                        (4097, 100, 1, 1), (70001, 100, 257, 10), (12000, 1536, 200, 10)])
@pytest.mark.parametrize("tile", ["128x128", "256x256"])
def test_mfma_exhaustive_scorer_equals_canonical_scan(func, n, dim, nq, k, tile):
    """The dense MFMA pass is only a filter; the answer must equal the canonical brute force
    bit for bit (ids, and distances from the canonical code) — with either block tile (the library picks 256 x 256 only for launches
    with thousands of tiles; the test knob forces each)."""
    import torch
    pg._lib.gpu_lib().hnsw_gpu_config_set(b"HNSW_GPU_BF_BIG_MIN_BLOCKS", b"0" if tile == "128x128" else b"-1")
    try:
        _mfma_case(func, n, dim, nq, k)
    finally:
        pg._lib.gpu_lib().hnsw_gpu_config_set(b"HNSW_GPU_BF_BIG_MIN_BLOCKS", None)


def test_mfma_tile_choice_of_a_large_launch_is_exact():
    """512 queries x 300 000 rows: the launch the library itself gives 256 x 256 tiles (an even number of 128-query tiles, > 2048 blocks)."""
    _mfma_case(pg.DIST_L2, 300_000, 32, 512, 10)


def _mfma_case(func, n, dim, nq, k):
    import torch
    X = gmm(n, dim, k=60, seed=19)
    X[100:140] = X[300:340]                  # exact duplicates: equal distances, tie by lower idx
    Q = gmm(nq, dim, k=60, seed=19, stream=1)
    Q[0] = X[5]                              # zero distance
    meta = pg.make_meta(min(dim, 1900), 4, 8, 8, func)
    ix = pg.GpuIndex.empty(meta, n)
    ix.append(X)
    dq = torch.from_numpy(Q).cuda()
    i0, d0 = ix.bruteforce_torch(dq, k)
    i1, d1 = ix.bruteforce_torch(dq, k, mfma=True)
    torch.cuda.synchronize()
    assert (i0 == i1).all()
    assert (d0.view(torch.int32) == d1.view(torch.int32)).all()
    ix.close()


def insert_paths():
    import ctypes as C
    paths = (C.c_uint64 * 2)()
    pg._lib.gpu_lib().hnsw_gpu_insert_path_counts(paths)
    return int(paths[0]), int(paths[1])


@pytest.mark.parametrize("func,dim,m,efc,fused", [(pg.DIST_L2, 24, 6, 40, "1"), (pg.DIST_COSINE, 100, 16, 64, "1"), (pg.DIST_MANHATTAN, 33, 5, 24, "1"),
                                                 (pg.DIST_L2, 24, 6, 40, "0"), (pg.DIST_L2, 16, 4, 600, "1"),
                                                 # lists longer than a wavefront (maxM = 80), bit matrices of 3 and 4 words per row, rows whose block needs > 48 KiB of LDS
                                                 (pg.DIST_L2, 8, 40, 100, "1"), (pg.DIST_COSINE, 8, 4, 188, "1"), (pg.DIST_L2, 16, 4, 230, "1"),
                                                 (pg.DIST_L2, 1536, 3, 12, "1")])
def test_insert_one_and_insert_candidates_build_the_oracles_graph(func, dim, m, efc, fused, monkeypatch):
    """hnsw_gpu_index_insert_one (append + serial link + changed lists in one call) and hnsw_gpu_index_insert_candidates (the same
    with the candidate list taken from a traced walk instead of a second search): row by row they build the graph the oracle's
    serial inserts build, byte for byte, and the lists they return are the lists the mirror holds."""
    import ctypes as C
    monkeypatch.setenv("HNSW_GPU_INSERT_FUSED", fused)
    before = insert_paths()
    n = 700 if efc < 100 and dim < 1000 else 400
    X = gmm(n, dim, k=20, seed=5 + dim)
    labels = np.arange(n, dtype=np.uint64) * 3 + 1
    port = oracle.PortIndex(dim, m, efc, 64, func)
    port.add(X, labels)
    meta = pg.make_meta(dim, m, efc, 64, func)
    maxM = int(meta.maxM)
    for use_candidates in (False, True):
        ix = pg.GpuIndex.empty(meta, n)
        mine = (C.c_uint32 * (maxM + 1))()
        others = (C.c_uint32 * (maxM * (maxM + 1)))()
        for i in range(n):
            p = np.ascontiguousarray(X[i])
            if use_candidates and i > 0:
                # what the validated cache does: a traced base-layer walk for the point on the mirror as it is, then the insert off its result
                ci, cd, pops, nev = ix.search_trace(p, efc, base=True)
                ci32 = np.ascontiguousarray(ci.astype(np.uint32))
                cd32 = np.ascontiguousarray(cd, dtype=np.float32)
                rc = ix.L.hnsw_gpu_index_insert_candidates(ix._h, p.ctypes.data, int(labels[i]), i, ci32.ctypes.data, cd32.ctypes.data,
                                                           len(ci32), mine, others)
            else:
                rc = ix.L.hnsw_gpu_index_insert_one(ix._h, p.ctypes.data, int(labels[i]), i, mine, others)
            assert rc == 0, (i, use_candidates)
            if i in (1, 50, n - 1):                              # the returned lists == the mirror's lists
                got = ix.export_flat().reshape(i + 1, -1)[:, :(maxM + 1) * 4].copy().view(np.uint32)
                assert (np.frombuffer(mine, np.uint32) [:1 + mine[0]] == got[i, :1 + mine[0]]).all()
                for j in range(mine[0]):
                    o = np.frombuffer(others, np.uint32)[j * (maxM + 1):(j + 1) * (maxM + 1)]
                    assert (o[:1 + o[0]] == got[mine[1 + j], :1 + o[0]]).all()
        got = ix.export_flat().reshape(n, -1)
        want = live_image(port.raw(), meta, n)
        assert (got == want).all(), f"use_candidates={use_candidates}: {(got != want).any(axis=1).sum()} elements differ"
        ix.close()
    # which path ran: two launches built for latency (device_insert.h) unless switched off or max(efConstruction, maxM + 1) = 600
    # candidates are more than its chain keeps in one wavefront (512) — then the general builder, same bytes
    after = insert_paths()
    two, general = after[0] - before[0], after[1] - before[1]
    assert (two, general) == ((2 * n, 0) if fused == "1" and efc <= 512 else (0, 2 * n)), (two, general)
