"""Fused search kernel vs the oracle on identical graph bytes: bit-exact ids, distances,
evaluation and hop counts; and vs the reference binary: ids equal except at near-ties."""
import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm, sift_like
from util import foreign_toolchain, REL_TOL, bits, build_port, classify_against_reference, mirror

pytestmark = pytest.mark.gpu

FUNCS = [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN]


def assert_same_as_oracle(ix, port, Q, ef):
    labels, dists, counts = ix.search(Q, ef)
    want = port.search_many(Q, ef)
    assert (counts == want["counts"]).all()
    for q in range(Q.shape[0]):
        c = counts[q]
        assert (labels[q, :c] == want["labels"][q, :c]).all(), f"query {q}: ids differ"
        assert (bits(dists[q, :c]) == bits(want["dists"][q, :c])).all(), f"query {q}: dists differ"
        assert (labels[q, c:] == pg.NO_LABEL).all() and np.isinf(dists[q, c:]).all()
    return labels, dists, counts


@pytest.mark.parametrize("func", FUNCS)
@pytest.mark.parametrize("dim,m,n", [(128, 8, 4000), (768, 16, 3000), (100, 4, 2000), (1536, 16, 1200)])
def test_search_bit_exact_vs_oracle(func, dim, m, n):
    port, X = build_port(n, dim, m, 48, func, seed=dim + func)
    Q = gmm(200, dim, k=50, seed=dim + func, stream=1)
    ix = mirror(port, func)
    for ef in (1, 10, 64, 128):
        assert_same_as_oracle(ix, port, Q, ef)
    ix.close()


def test_stats_match_oracle_counters():
    """E_q / H_q reported by the kernel == coords / link reads of the oracle."""
    import torch
    port, X = build_port(5000, 128, 8, 64, pg.DIST_L2, seed=11)
    Q = gmm(300, 128, k=50, seed=11, stream=1)
    ix = mirror(port, pg.DIST_L2)
    out = ix.search_torch(torch.from_numpy(Q).cuda(), 128, stats=True)
    torch.cuda.synchronize()
    want = port.search_many(Q, 128)
    st = out["stats"].cpu().numpy().astype(np.uint32)
    assert (st[:, 0] == want["evals"]).all()
    assert (st[:, 1] == want["hops"]).all()
    assert (out["labels"].cpu().numpy().view(np.uint64) == want["labels"]).all()
    assert ix.last_search_ms() > 0
    base = ix.search_torch(torch.from_numpy(Q).cuda(), 64, base=True)
    torch.cuda.synchronize()
    for q in range(0, 300, 17):
        idx, d, _, _ = port.search_base(Q[q], 64)
        c = int(base["counts"][q])
        assert c == idx.size
        assert (base["idx"][q, :c].cpu().numpy().view(np.uint32) == idx).all()
    ix.close()


@pytest.mark.parametrize("func", FUNCS)
def test_search_vs_reference_binary(func):
    """Same graph bytes, reference CPU search vs device: ids identical wherever the
    reference's own neighbouring distances are not within 1e-5 (north-star contract)."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    dim, m, n = 256, 12, 4000
    X = gmm(n, dim, k=60, seed=5 + func)
    ref = oracle.RefIndex(dim, m, 64, 100, func)
    ref.add(X)
    Q = gmm(600, dim, k=60, seed=5 + func, stream=2)
    want = ref.search_many(Q, 100)
    ix = mirror(ref, func)
    labels, dists, counts = ix.search(Q, 100)
    assert (counts == want["counts"]).all()
    # device == oracle bit for bit on these bytes; oracle vs reference: every query classified — ids equal
    # unless a decision of the walk compared two distances within REL_TOL (1e-5, not a multiple of it)
    # and the reference's summation order flips it (util.classify_against_reference)
    port = oracle.PortIndex(dim, m, 64, 100, func, capacity=n)
    port.load_raw(ref.raw(), n)
    got = classify_against_reference(port, Q, 100, want["labels"])
    assert (labels == got["labels"]).all() and (bits(dists) == bits(got["dists"])).all()
    c = got["classification"]
    assert c["mismatch_unexplained"] == 0 and c["identical_ids"] >= 0.9 * Q.shape[0]
    ix.close()


def test_toy_golden_orderings():
    """knn.out:16-19,39-42,56-59 restated at the C boundary (labels = insertion order)."""
    rows = np.array([[0, 1, 2], [1, 2, 3], [1, 1, 1], [1, 2, 4]], np.float32)
    q = np.array([[3, 3, 3]], np.float32)
    expect = {pg.DIST_L2: [1, 3, 2, 0], pg.DIST_COSINE: [2, 1, 3, 0], pg.DIST_MANHATTAN: [1, 3, 0, 2]}
    for func, order in expect.items():
        port = oracle.PortIndex(3, 3, 5, 5, func)
        port.add(rows)
        ix = mirror(port, func)
        labels, dists, counts = ix.search(q, 5)
        assert counts[0] == 4 and labels[0, :4].tolist() == order
        ix.close()


def test_empty_index_returns_no_rows():
    """gh-2.out:5-8 / hnswalg.cpp:56-57."""
    meta = pg.make_meta(3, 3, 5, 5, pg.DIST_L2)
    ix = pg.GpuIndex.from_flat(meta, np.zeros(0, np.uint8), 0)
    labels, dists, counts = ix.search(np.zeros((4, 3), np.float32), 5)
    assert (counts == 0).all() and (labels == pg.NO_LABEL).all()
    ix.close()


def test_single_element_and_ef_larger_than_index():
    port, X = build_port(37, 16, 4, 16, pg.DIST_L2, k=5, seed=2)
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(20, 16, k=5, seed=2, stream=3)
    labels, dists, counts = assert_same_as_oracle(ix, port, Q, 128)
    assert (counts == 37).all()
    ix.close()
    one = oracle.PortIndex(16, 4, 16, 8, pg.DIST_L2)
    one.add(X[:1], np.array([77], np.uint64))
    ix = mirror(one, pg.DIST_L2)
    labels, dists, counts = ix.search(Q, 8)
    assert (counts == 1).all() and (labels[:, 0] == 77).all()
    ix.close()


def test_deleted_labels_are_filtered_after_search():
    """knn.out:93-122: vacuumed rows are traversed but not returned (hnswalg.cpp:245)."""
    port, X = build_port(1500, 64, 6, 32, pg.DIST_L2, seed=9)
    rng = np.random.default_rng(9)
    dead = rng.choice(1500, 400, replace=False)
    for i in dead:
        port.set_deleted(int(i))
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(100, 64, k=50, seed=9, stream=1)
    labels, dists, counts = assert_same_as_oracle(ix, port, Q, 64)
    assert (counts < 64).any()
    live = labels[labels != pg.NO_LABEL]
    assert not ((live >> np.uint64(48)) & np.uint64(1)).any()
    # flag set after the mirror was built
    ix.set_deleted(int(labels[0, 0]) & 0xFFFFFFFF)
    port.set_deleted(int(labels[0, 0]) & 0xFFFFFFFF)
    assert_same_as_oracle(ix, port, Q, 64)
    ix.close()


@pytest.mark.parametrize("env", [{}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}, {"HNSW_GPU_TEAM": "0"}])
@pytest.mark.parametrize("dim", [64, 768])
def test_search_trace_reports_the_walk(dim, env, monkeypatch):
    """hnsw_gpu_search_trace: results as hnsw_search gives them plus the walk's pop sequence (hnswalg.cpp:73) — equal, element
    for element, to the oracle's; in every kernel form (beam / team, two-set registers, LDS arrays), for hnsw_search
    and for searchBaseLayer alone.  This sequence is what the drop-in library validates against the host's pages."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    port, X = build_port(5000, dim, 8, 40, pg.DIST_L2, seed=31 + dim)
    for i in (3, 77, 400):
        port.set_deleted(i)
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(24, dim, k=50, seed=31 + dim, stream=1)
    for ef in (10, 64, 300):
        for q in Q:
            for base in (False, True):
                lab, dst, pops, ev = ix.search_trace(q, ef, base=base)
                wl, wd, wp, wev = port.search_trace(q, ef, base=base)
                assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
                assert len(pops) == len(wp) and (pops == wp).all() and ev == wev
    # a capacity smaller than the walk: the count is still the walk's, the stored prefix is its beginning
    lab, dst, pops, ev = ix.search_trace(Q[0], 64, pops_cap=5)
    assert (pops == port.search_trace(Q[0], 64)[2][:5]).all()
    # the three-step form: pops handed out while the kernel runs (system-scope stores into pinned host memory)
    for q in Q[:8]:
        lab, dst, pops, ev, polls = ix.search_trace_polled(q, 64)
        wl, wd, wp, wev = port.search_trace(q, 64)
        assert (lab == wl).all() and (bits(dst) == bits(wd)).all() and len(pops) == len(wp) and (pops == wp).all() and ev == wev
        assert polls >= len(wp) // 5
    # a trace that is begun and never ended (its caller was thrown out of its own code) does not disturb the next one
    ix.L.hnsw_gpu_search_trace_begin(ix._h, np.ascontiguousarray(Q[1]).ctypes.data, 64, 0, 1 << 10)
    lab, dst, pops, ev = ix.search_trace(Q[2], 64)
    assert (pops == port.search_trace(Q[2], 64)[2]).all()
    ix.close()


def test_vacuum_flags_in_one_batch():
    """hnsw_gpu_index_set_deleted_batch: what a VACUUM does to many rows (embedding.c:883-946), set and cleared again,
    with a repeated element in the list; results track the oracle's flags."""
    port, X = build_port(1200, 32, 6, 32, pg.DIST_L2, seed=19)
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(80, 32, k=50, seed=19, stream=1)
    rng = np.random.default_rng(19)
    dead = rng.choice(1200, 500, replace=False).astype(np.uint32)
    ix.set_deleted_many(np.concatenate([dead, dead[:7]]))
    for i in dead:
        port.set_deleted(int(i))
    labels, _, counts = assert_same_as_oracle(ix, port, Q, 48)
    assert (counts < 48).any()
    ix.set_deleted_many(dead[:250], deleted=False)
    for i in dead[:250]:
        port.set_deleted(int(i), False)
    assert_same_as_oracle(ix, port, Q, 48)
    with pytest.raises(Exception):
        ix.set_deleted_many(np.array([5, 1200], np.uint32))      # past the end: refused, nothing half-applied to check here
    ix.close()


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_MANHATTAN])
def test_exact_ties_follow_the_pair_order(func):
    """Integer data and duplicated rows: distances tie exactly, so results are decided by the
    (dist, idx) / (dist, label) pair comparisons of the reference heaps (hnswalg.cpp:52-53,236)."""
    X = sift_like(2500, 32, k=20, seed=4)
    X[1000:1400] = X[200:600]                    # exact duplicates
    port = oracle.PortIndex(32, 6, 40, 64, func)
    labels_in = np.arange(2500, dtype=np.uint64)[::-1].copy()    # label order != idx order
    port.add(X, labels_in)
    ix = mirror(port, func)
    Q = np.concatenate([X[200:260], sift_like(60, 32, k=20, seed=4, stream=1)])
    labels, dists, counts = assert_same_as_oracle(ix, port, Q, 100)
    d = dists[:, :100]
    assert (np.diff(d, axis=1) == 0).any(), "test data produced no ties"
    if oracle.have_ref():
        ref = oracle.RefIndex(32, 6, 40, 64, func)
        ref.load_raw(port.raw(), 2500)
        want = ref.search_many(Q, 100)
        assert (labels == want["labels"]).all()      # integer data: bit-exact vs the reference too
    ix.close()


@pytest.mark.parametrize("ef", [200, 512, 1500, 2800, 8192, 65536])
def test_large_ef(ef):
    """efSearch doubling path of the scan (embedding.c:334) reaches large beams — up to "more than the
    index holds".  Past what LDS can hold at 4 waves per CU the result/candidate arrays live in HBM."""
    port, X = build_port(3000, 48, 8, 32, pg.DIST_L2, seed=13)
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(40 if ef <= 1500 else 8, 48, k=50, seed=13, stream=1)
    assert_same_as_oracle(ix, port, Q, ef)
    ix.close()


@pytest.mark.parametrize("func", FUNCS)
def test_generic_form_with_sets_in_hbm_is_exact(func, monkeypatch):
    """Force the HBM-resident arrays at a moderate ef (vacuumed labels and equal distances included)."""
    monkeypatch.setenv("HNSW_GPU_LDS_SET_MIN_WAVES", "1000")
    rng = np.random.default_rng(5)
    X = np.rint(gmm(4000, 20, k=40, seed=9) * 3).astype(np.float32)
    if func == pg.DIST_COSINE:
        X[(X * X).sum(axis=1) == 0] = 1.0
    port = oracle.PortIndex(20, 6, 32, 64, func)
    port.add(X)
    for i in rng.choice(4000, 500, replace=False):
        port.set_deleted(int(i))
    ix = mirror(port, func)
    Q = gmm(24, 20, k=40, seed=9, stream=1)
    if func == pg.DIST_COSINE:
        Q[(Q * Q).sum(axis=1) == 0] = 1.0
    for ef in (300, 700):
        assert_same_as_oracle(ix, port, Q, ef)
    ix.close()


def test_default_m_100_link_lists_longer_than_a_wave():
    """DEFAULT_M = 100 -> maxM = 200 links per element (embedding.c:113,224)."""
    port, X = build_port(1200, 24, 100, 16, pg.DIST_COSINE, seed=21)
    ix = mirror(port, pg.DIST_COSINE)
    Q = gmm(50, 24, k=50, seed=21, stream=1)
    assert_same_as_oracle(ix, port, Q, 64)
    ix.close()


def test_export_roundtrip_is_identity():
    port, X = build_port(2000, 40, 5, 32, pg.DIST_L2, seed=17)
    ix = mirror(port, pg.DIST_L2)
    meta = ix.meta
    got = ix.export_flat().reshape(2000, -1)
    want = port.raw().reshape(2000, -1).copy()
    # slots past `count` are dead bytes in a host image (a pruned list keeps its old tail,
    # hnswalg.cpp:214-219); the mirror writes zeros there
    lw = want[:, :meta.offset_data].copy().view(np.uint32)
    for e in range(2000):
        lw[e, 1 + lw[e, 0]:] = 0
    want[:, :meta.offset_data] = lw.view(np.uint8)
    assert (got == want).all()
    again = pg.GpuIndex.from_flat(meta, got.ravel(), 2000)
    assert (again.export_flat().reshape(2000, -1) == got).all()
    again.close()
    ix.close()


def test_append_then_export_matches_zero_linked_elements():
    meta = pg.make_meta(10, 4, 8, 8, pg.DIST_L2)
    ix = pg.GpuIndex.empty(meta, 100)
    X = gmm(60, 10, k=5, seed=1)
    lab = np.arange(100, 160, dtype=np.uint64)
    ix.append(X[:25], lab[:25])
    ix.append(X[25:], lab[25:])
    raw = ix.export_flat().reshape(60, -1)
    esz = meta.size_data_per_element
    assert raw.shape[1] == esz
    assert (raw[:, :meta.offset_data] == 0).all()
    assert (raw[:, meta.offset_data:meta.offset_label].copy().view(np.float32) == X).all()
    assert (raw[:, meta.offset_label:].copy().view(np.uint64).ravel() == lab).all()
    ix.close()


def test_corrupt_image_is_rejected():
    port, X = build_port(50, 8, 3, 8, pg.DIST_L2, k=3, seed=1)
    raw = port.raw().reshape(50, -1).copy()
    raw[7, 4:8] = np.frombuffer(np.uint32(9999).tobytes(), np.uint8)     # link to a missing element
    meta = pg.make_meta(8, 3, 8, 8, pg.DIST_L2)
    with pytest.raises(RuntimeError, match="corrupt"):
        pg.GpuIndex.from_flat(meta, raw.ravel(), 50)


def test_many_queries_reuse_slots():
    """More queries than resident waves: every slot runs several queries, so the visited
    bitmap must come back clean each time."""
    port, X = build_port(6000, 64, 8, 48, pg.DIST_L2, seed=23)
    ix = mirror(port, pg.DIST_L2)
    Q = gmm(30000, 64, k=50, seed=23, stream=1)
    labels, dists, counts = ix.search(Q, 32)
    sel = np.random.default_rng(0).choice(30000, 400, replace=False)
    want = port.search_many(Q[sel], 32)
    assert (labels[sel] == want["labels"]).all()
    labels2, _, _ = ix.search(Q, 32)                # second launch on the same workspace
    assert (labels2 == labels).all()
    ix.close()


@pytest.mark.parametrize("env", [
    {"HNSW_GPU_HASH_ENTRIES": "512"},        # tiny LDS visited set: most ids spill to the HBM bitmap
    {"HNSW_GPU_HASH_ENTRIES": "0"},          # bitmap only
    {"HNSW_GPU_FORCE_LDS_HEAPS": "1"},       # generic kernel (sorted arrays in LDS) at small ef
    {"HNSW_GPU_BEAM": "0"},                  # no beam form: the generic kernel (the two-set register form lives in experiment builds only)
    {"HNSW_GPU_BEAM": "0", "HNSW_GPU_HASH_ENTRIES": "512"},
    {"HNSW_GPU_BEAM16": "0"},                # ef in (256, 512]: LDS form instead of 16 set registers
    {"HNSW_GPU_TEAM": "1"},                  # team form: idle waves of a block feed a sibling's walk from LDS caches
    {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "4"},
    {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "2", "HNSW_GPU_HASH_ENTRIES": "512"},
    {"HNSW_GPU_TEAM": "0"},
    {},
])
def test_every_kernel_variant_is_exact(env, monkeypatch):
    """The visited set may live in LDS, spill to the bitmap half-way through a query, or be the
    bitmap alone; the accepted set may be one counted set in registers (beam form) or two unsorted
    arrays in LDS (generic form): all must give the oracle's answer."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    port, X = build_port(20000, 768, 16, 64, pg.DIST_L2, k=100, seed=77)
    Q = gmm(300, 768, k=100, seed=77, stream=1)
    ix = mirror(port, pg.DIST_L2)
    for ef in (40, 100, 256, 400):
        import torch
        out = ix.search_torch(torch.from_numpy(Q).cuda(), ef, stats=True)
        torch.cuda.synchronize()
        want = port.search_many(Q, ef)
        assert (out["labels"].cpu().numpy().view(np.uint64) == want["labels"]).all()
        assert (bits(out["dists"].cpu().numpy()) == bits(want["dists"])).all()
        st = out["stats"].cpu().numpy().astype(np.uint32)
        assert (st[:, 0] == want["evals"]).all() and (st[:, 1] == want["hops"]).all()
    if env.get("HNSW_GPU_HASH_ENTRIES") == "512":
        assert want["evals"].max() > 512          # the spill path really ran
    # a second launch on the same slots: bitmap bits set by the spill path were undone
    labels2, _, _ = ix.search(Q, 100)
    assert (labels2 == port.search_many(Q, 100)["labels"]).all()
    ix.close()


@pytest.mark.parametrize("env", [{}, {"HNSW_GPU_HASH_ENTRIES": "512"}, {"HNSW_GPU_NARROW5": "0"}])
@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_MANHATTAN])
@pytest.mark.parametrize("dim", [128, 72])
def test_narrow_rows_five_waves_form_is_exact(dim, func, env, monkeypatch):
    """Rows of up to 128 floats, ef <= 128, launches too large for the team form: the beam kernel runs the 8-rows-per-pass
    shape at 5 waves/SIMD with a bucketed visited set whose bucket count is not a power of two (432 at the default LDS
    budget; with 128 buckets and E_q of 300-600 ids per query a bucket of eight overflows into the bitmap in most queries).
    ids, distance bits, E_q, H_q = the oracle's; the same with the form switched off (HNSW_GPU_NARROW5=0: 16 rows per
    pass, 4 waves/SIMD)."""
    import torch
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    port, X = build_port(12000, dim, 16, 64, func, k=60, seed=500 + dim)
    Q = gmm(900, dim, k=60, seed=500 + dim, stream=1)
    ix = mirror(port, func)
    dQ = torch.from_numpy(Q).cuda()
    for ef in (48, 128):
        want = port.search_many(Q, ef, nthreads=8)
        for rep in range(2):                 # second launch: the bitmap bits of overflowing buckets were undone
            out = ix.search_torch(dQ, ef, stats=True)
            torch.cuda.synchronize()
            assert ("Shape2x2" in ix.last_search_kernel()) == (env.get("HNSW_GPU_NARROW5") != "0")
            assert (out["labels"].cpu().numpy().view(np.uint64) == want["labels"]).all()
            assert (bits(out["dists"].cpu().numpy()) == bits(want["dists"])).all()
            st = out["stats"].cpu().numpy().astype(np.uint32)
            assert (st[:, 0] == want["evals"]).all() and (st[:, 1] == want["hops"]).all()
    ix.close()


@pytest.mark.parametrize("func", FUNCS)
@pytest.mark.parametrize("dim,m", [(128, 16), (768, 16), (1536, 32), (100, 40)])
def test_team_form_is_exact_at_every_launch_size(func, dim, m, monkeypatch):
    """Team form (device_search.h): from one query per launch (1 walking wave + 7 helpers per block) through
    launches where most waves walk and helpers only appear in the tail — ids, distance bits, E_q and H_q must
    equal the oracle's, i.e. the one-wave form's.  m=40 exercises link lists longer than one wave (maxM=80)."""
    import torch
    monkeypatch.setenv("HNSW_GPU_TEAM", "1")
    n = 6000
    port, X = build_port(n, dim, m, 48, func, seed=3 * dim + func)
    Q = gmm(700, dim, k=50, seed=3 * dim + func, stream=1)
    ix = mirror(port, func)
    ef = 96
    want = port.search_many(Q, ef, nthreads=8)
    dQ = torch.from_numpy(Q).cuda()
    for nq in (1, 3, 64, 700):
        for rep in range(2):               # second launch: caches and control words start from a used LDS
            out = ix.search_torch(dQ[:nq].contiguous(), ef, stats=True)
            torch.cuda.synchronize()
            assert ", true, " in ix.last_search_kernel()          # (the TEAM template argument)
            assert (out["labels"].cpu().numpy().view(np.uint64) == want["labels"][:nq]).all(), (nq, rep)
            assert (bits(out["dists"].cpu().numpy()) == bits(want["dists"][:nq])).all()
            st = out["stats"].cpu().numpy().astype(np.uint32)
            assert (st[:, 0] == want["evals"][:nq]).all() and (st[:, 1] == want["hops"][:nq]).all()
    ix.close()


@pytest.mark.parametrize("dim,m", [(768, 16), (128, 16)])
def test_walkers_per_block_of_a_small_launch_change_nothing_but_its_shape(dim, m):
    """hnsw_gpu_ctx_set_walkers (the batching server's load policy): a small team launch with 1 .. 8 walking waves per block —
    from every walk with seven helpers to every wave walking — answers every query alike: labels, distance bits, E_q, H_q equal the
    oracle's, with several such launches in flight on different streams too."""
    import torch
    func, n, ef = pg.DIST_L2, 8000, 96
    port, X = build_port(n, dim, m, 48, func, k=40, seed=11 * dim)
    Q = gmm(300, dim, k=40, seed=11 * dim, stream=1)
    want = port.search_many(Q, ef, nthreads=8)
    ix = mirror(port, func)
    dQ = torch.from_numpy(Q).cuda()
    ctxs = [pg.SearchContext(ix) for _ in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]
    outs = [ix.search_torch(dQ, ef, stats=True) for _ in range(3)]
    torch.cuda.synchronize()
    for walkers in (0, 1, 2, 3, 8):
        for nq in (5, 190, 300):
            for k in range(3):
                ctxs[k].set_walkers(walkers)
                ctxs[k].search_torch(dQ[:nq].contiguous(), ef, outs[k], streams[k])
            torch.cuda.synchronize()
            for k in range(3):
                assert (outs[k]["labels"][:nq].cpu().numpy().view(np.uint64) == want["labels"][:nq]).all(), (walkers, nq, k)
                assert (bits(outs[k]["dists"][:nq].cpu().numpy()) == bits(want["dists"][:nq])).all(), (walkers, nq, k)
                st = outs[k]["stats"][:nq].cpu().numpy().astype(np.uint32)
                assert (st[:, 0] == want["evals"][:nq]).all() and (st[:, 1] == want["hops"][:nq]).all(), (walkers, nq, k)
    for c in ctxs:
        c.close()
    ix.close()


@pytest.mark.timeout(180, method="thread")
@pytest.mark.parametrize("dim,m,func,walkers", [(768, 16, pg.DIST_L2, 4), (128, 16, pg.DIST_L2, 8), (100, 8, pg.DIST_COSINE, 1), (1536, 32, pg.DIST_COSINE, 2)])
def test_a_stream_answers_like_a_launch(dim, m, func, walkers):
    """Streams (include/hnsw_gpu.h): ONE resident launch fed from the host while it runs.  Queries published a few at a time while
    others are still walking, a ring far smaller than the number of queries (every slot reused many times), the stream closed and
    another opened: every answer — labels, distance bits, counts — equals the oracle's, i.e. a plain launch's; the mirror's ordinary
    launches work before, between and after."""
    import torch
    n, ef, nq, ring = 8000, 96, 1500, 64
    port, X = build_port(n, dim, m, 48, func, k=40, seed=17 * dim + walkers)
    Q = gmm(nq, dim, k=40, seed=17 * dim + walkers, stream=1)
    want = port.search_many(Q, ef, nthreads=8)
    ix = mirror(port, func)
    assert_same_as_oracle(ix, port, Q[:40], ef)
    ctx = pg.SearchContext(ix)
    for round_ in range(2):
        # round 0: the default completion form (results as system-scope stores, the flag behind vmcnt(0)); round 1: plain stores and a
        # full system-scope release per answered query (HNSW_GPU_STREAM_LIGHT)
        pg.config_set("HNSW_GPU_STREAM_LIGHT", None if round_ == 0 else 0)
        st = pg.SearchStream(ctx, ef, ring=ring, walkers=walkers)
        try:
            _drive_stream(st, Q, want, nq, ring, round_)
        finally:
            pg.config_set("HNSW_GPU_STREAM_LIGHT", None)
            st.close()                                          # (a resident launch left behind holds the whole device: everything after it would wait)
        assert_same_as_oracle(ix, port, Q[40:80], ef)            # an ordinary launch on the mirror after the stream has left
    ctx.close()
    ix.close()


def test_a_plain_launch_issued_while_a_stream_is_open_runs_when_the_stream_lets_go(capsys):
    """A stream's resident launch is sized to exactly the blocks the device holds (wide rows: 2 waves/SIMD, the register file is
    full), so an ordinary launch on the same device has nowhere to run while the stream is open: it is queued BEHIND the stream — not
    beside it, not lost — and runs the moment the stream closes; the stream keeps answering meanwhile.  That is why a stream owns its
    device (include/hnsw_gpu.h "Streams", DESIGN.md 4.6: the server closes a session for writers, control work and after 2 ms without
    searches).  The test states what happens with its numbers; both sides' answers must equal the oracle's either way."""
    import time
    import torch
    n, dim, m, ef, nq = 20000, 768, 16, 64, 20000
    port, X = build_port(n, dim, m, 48, pg.DIST_L2, k=40, seed=91)
    Q = gmm(nq, dim, k=40, seed=91, stream=1)
    want = port.search_many(Q[:512], ef, nthreads=8)
    ix = mirror(port, pg.DIST_L2)
    dq = torch.from_numpy(Q).cuda()
    out = ix.search_torch(dq, ef)                              # (buffers; also the time of the launch on a free device)
    torch.cuda.synchronize()
    t_free = ix.last_search_ms()
    ctx_s, ctx_p = pg.SearchContext(ix), pg.SearchContext(ix)
    side = torch.cuda.Stream()
    st = pg.SearchStream(ctx_s, ef, ring=256)
    try:
        slots = st.submit(Q[:64])                              # the stream is alive and answering
        lab, dst, cnt = st.wait(slots)
        assert (lab == want["labels"][:64]).all() and (bits(dst) == bits(want["dists"][:64])).all()
        out["labels"].zero_()
        ev = torch.cuda.Event()
        t0 = time.perf_counter()
        ctx_p.search_torch(dq, ef, out, stream=side)           # an ordinary launch of 20 000 queries, same device, another stream
        ev.record(side)
        beside = False
        answered_meanwhile = 0
        while time.perf_counter() - t0 < 1.0:                  # one second: forty times what the launch takes on a free device
            if ev.query():
                beside = True
                break
            slots = st.submit(Q[64 + answered_meanwhile:64 + answered_meanwhile + 32])
            lab, dst, cnt = st.wait(slots)                     # the open stream still answers while that launch waits
            assert (lab == want["labels"][64 + answered_meanwhile:96 + answered_meanwhile]).all()
            answered_meanwhile = (answered_meanwhile + 32) % 384
        t_waited = time.perf_counter() - t0
    finally:
        t1 = time.perf_counter()
        st.close()
    ev.synchronize()
    t_after_close = time.perf_counter() - t1
    got = out["labels"][:512].cpu().numpy()
    assert (got == want["labels"]).all()                       # the queued launch ran, whole and exact
    with capsys.disabled():
        print(f"\n[stream + plain launch on one device] 20 000-query launch: {t_free:.2f} ms on a free device; issued while a stream was open it "
              f"{'ran BESIDE the resident launch and ended after %.1f ms' % (t_waited * 1e3) if beside else 'had not ended after %.0f ms (the stream answered queries meanwhile)' % (t_waited * 1e3)}"
              f"; it ended {t_after_close * 1e3:.1f} ms after the stream was closed")
    ctx_s.close(); ctx_p.close(); ix.close()


def _drive_stream(st, Q, want, nq, ring, round_):
    assert st.alive()
    done = 0
    rng = np.random.default_rng(round_)
    pending = []                                             # (first query number, slots)
    while done < nq or pending:
        # keep up to ring/2 queries in flight, published in irregular chunks
        inflight = sum(len(sl) for _, sl in pending)
        if done < nq and inflight <= ring // 2:
            k = int(min(nq - done, rng.integers(1, ring // 2 - 1), ring - inflight - 1))
            pending.append((done, st.submit(Q[done:done + k])))
            done += k
            continue
        first, slots = pending.pop(0)
        lab, dst, cnt = st.wait(slots)
        k = len(slots)
        assert (cnt == want["counts"][first:first + k]).all(), (round_, first)
        assert (lab == want["labels"][first:first + k]).all(), (round_, first)
        assert (bits(dst) == bits(want["dists"][first:first + k])).all(), (round_, first)


@pytest.mark.parametrize("ef", [1, 5, 64, 128, 256])
def test_beam_prune_with_ties_at_the_bound(ef):
    """0/1 vectors in 6 dimensions: only 7 distinct L2 distances, so the ef-th smallest distance is
    shared by hundreds of elements.  The beam form prunes its accepted set at that distance and must
    keep every tie (they stay live candidates), then pick the survivors by (dist, idx)."""
    rng = np.random.default_rng(11)
    X = rng.integers(0, 2, size=(3000, 6)).astype(np.float32)
    Q = rng.integers(0, 2, size=(64, 6)).astype(np.float32)
    port = oracle.PortIndex(6, 8, 40, 64, pg.DIST_L2)
    port.add(X)
    ix = mirror(port, pg.DIST_L2)
    assert_same_as_oracle(ix, port, Q, ef)
    ix.close()


def test_visited_sets_larger_than_the_lds_table():
    """Weakly clustered data at ef=256: thousands of evaluations per query (> 3072-entry budget)."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((40000, 24)).astype(np.float32)
    Q = rng.standard_normal((100, 24)).astype(np.float32)
    port = oracle.PortIndex(24, 16, 40, 256, pg.DIST_L2)
    port.add(X)
    ix = mirror(port, pg.DIST_L2)
    labels, dists, counts = assert_same_as_oracle(ix, port, Q, 256)
    assert port.search_many(Q, 256)["evals"].max() > 3100
    ix.close()


def test_incremental_mirror_update_from_changed_elements():
    """A host that inserts on the CPU pushes only the changed elements (the new ones and the
    neighbours whose link lists were rewritten) instead of re-mirroring everything."""
    dim, m, n0, n1 = 40, 6, 3000, 3400
    X = gmm(n1, dim, k=30, seed=61)
    port = oracle.PortIndex(dim, m, 32, 64, pg.DIST_L2)
    port.add(X[:n0])
    ix = mirror(port, pg.DIST_L2)
    before = port.raw().reshape(n0, -1).copy()
    port.add(X[n0:])
    after = port.raw().reshape(n1, -1)
    changed = np.nonzero((after[:n0] != before).any(axis=1))[0]
    assert 0 < changed.size < n0                       # only some old elements were re-linked
    ix.update_from_flat(after[n0:].ravel(), n0, n1 - n0)          # new elements (grows the mirror)
    for e in changed:                                             # re-linked neighbours, one by one
        ix.update_from_flat(after[e].ravel(), int(e), 1)
    port.set_deleted(5)
    ix.update_from_flat(port.raw().reshape(n1, -1)[5].ravel(), 5, 1)   # a vacuum flag is an element change too
    Q = gmm(150, dim, k=30, seed=61, stream=1)
    assert_same_as_oracle(ix, port, Q, 64)
    with pytest.raises(RuntimeError, match="gap"):
        ix.update_from_flat(after[0].ravel(), n1 + 5, 1)
    ix.close()


def test_search_contexts_overlap_on_two_streams():
    """Two contexts on two streams, interleaved launches: every launch gives the oracle's answer."""
    import torch
    port, X = build_port(8000, 96, 8, 48, pg.DIST_L2, seed=88)
    ix = mirror(port, pg.DIST_L2)
    Qa = torch.from_numpy(gmm(3000, 96, k=50, seed=88, stream=1)).cuda()
    Qb = torch.from_numpy(gmm(3000, 96, k=50, seed=88, stream=2)).cuda()
    ctx = [pg.SearchContext(ix), pg.SearchContext(ix)]
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = [ix.search_torch(Qa, 64), ix.search_torch(Qb, 64)]
    torch.cuda.synchronize()
    for o in outs:
        o["labels"].zero_()
    for rep in range(4):
        ctx[0].search_torch(Qa, 64, outs[0], st[0])
        ctx[1].search_torch(Qb, 64, outs[1], st[1])
    torch.cuda.synchronize()
    wa = port.search_many(Qa.cpu().numpy(), 64)
    wb = port.search_many(Qb.cpu().numpy(), 64)
    assert (outs[0]["labels"].cpu().numpy().view(np.uint64) == wa["labels"]).all()
    assert (outs[1]["labels"].cpu().numpy().view(np.uint64) == wb["labels"]).all()
    for c in ctx:
        c.close()
    ix.close()


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
def test_streamed_completion_and_host_context_forms(func):
    """hnsw_gpu_search_batch_ctx_flags: the kernel writes results and per-query completion flags
    straight into pinned host memory; hnsw_gpu_search_batch_ctx_host: host arrays on the context's own
    stream.  Same bits as the oracle in every kernel form (beam, beam with 16 registers / LDS, generic)."""
    port, X = build_port(6000, 72, 6, 40, func, seed=120 + func)
    ix = mirror(port, func)
    ctx = pg.SearchContext(ix)
    Q = gmm(700, 72, k=50, seed=120 + func, stream=1)
    for ef in (48, 300, 700):
        want = port.search_many(Q, ef)
        lab, dst, cnt, order = ctx.search_streamed(Q, ef)
        assert sorted(order.tolist()) == list(range(len(Q)))           # every flag arrived, once
        assert (cnt == want["counts"]).all()
        for q in range(len(Q)):
            k = int(cnt[q])
            assert (lab[q, :k] == want["labels"][q, :k]).all()
            assert (bits(dst[q, :k]) == bits(want["dists"][q, :k])).all()
        lab2, dst2, cnt2 = ctx.search_host(Q, ef)
        assert (lab2 == lab).all() and (bits(dst2) == bits(dst)).all() and (cnt2 == cnt).all()
    ctx.close()
    ix.close()


def test_nan_distances_do_not_hang_or_crash():
    """Zero vectors give NaN cosine distances in the reference too (distfunc.c:144, 0/0); they are
    outside the parity contract but must not hang the kernel or corrupt memory."""
    X = gmm(3000, 32, k=10, seed=9)
    X[::7] = 0.0
    port = oracle.PortIndex(32, 6, 24, 32, pg.DIST_COSINE)
    port.add(X[:40])                         # a few inserts through the oracle (NaNs included) ...
    meta = pg.make_meta(32, 6, 24, 32, pg.DIST_COSINE)
    ix = pg.GpuIndex.empty(meta, 3000)
    ix.append(X)
    ix.link(0, 3000)                         # ... and the whole set through the device builder
    Q = gmm(500, 32, k=10, seed=9, stream=1)
    Q[::5] = 0.0
    for ef in (16, 200, 400):
        labels, dists, counts = ix.search(Q, ef)
        assert (counts <= ef).all()
        ok = labels != pg.NO_LABEL
        assert (labels[ok] < 3000).all()
    raw = ix.export_flat().reshape(3000, -1)
    cnt = raw[:, :4].copy().view(np.uint32).ravel()
    assert cnt.max() <= 12
    ix.close()


@pytest.mark.parametrize("env", [{}, {"HNSW_GPU_TEAM": "0"}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}])
def test_evaluation_trace_is_the_walk_and_replay_reads_its_bytes(env, monkeypatch):
    """Measurement entry points of bench.py's replay roof (include/hnsw_gpu.h): the traced rows of every query are exactly what its
    walk must score (entry point, then the unvisited links of every popped element, in order: restated from the pop sequence and
    the element image), the per-query clock stamps are ordered, and the replay gathers exactly the traced bytes."""
    import torch
    from util import evals_from_pops
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, dim, m, ef = 8000, 200, 16, 100
    port, X = build_port(n, dim, m, 48, pg.DIST_L2, seed=31)
    Q = gmm(300, dim, k=50, seed=31, stream=1)
    ix = mirror(port, pg.DIST_L2)
    tr = ix.search_traced_torch(torch.from_numpy(Q).cuda(), ef, evals_cap=2048)
    torch.cuda.synchronize()
    slots = ix.last_search_slots()
    want = port.search_many(Q, ef, nthreads=8)
    st = tr["stats"].cpu().numpy().astype(np.uint32)
    assert (tr["labels"].cpu().numpy().view(np.uint64) == want["labels"]).all()
    assert (st[:, 0] == want["evals"]).all() and (st[:, 1] == want["hops"]).all()
    ev = tr["evals"].cpu().numpy().view(np.uint32)
    tm = tr["times"].cpu().numpy()
    assert (tm[:, 1] >= tm[:, 0]).all() and (tm[:, 0] > 0).all()
    for i in range(0, 300, 23):
        _, _, pops, nev = ix.search_trace(Q[i], ef)
        w = evals_from_pops(port.raw(), ix.meta, n, ix.meta.enterpoint_node, pops)
        assert nev == len(w) == st[i, 0] and (ev[i, :len(w)] == w).all(), i
    # the replay reads exactly the traced rows, whole: bytes, and the (order-free) sum of the bit patterns of every word
    words = X.view(np.uint32).astype(np.uint64).sum(axis=1)
    want_sum = int(sum(int(words[ev[i, :st[i, 0]]].sum()) for i in range(300)) % (1 << 64))
    for kb, rpg in ((4, 2), (4, 4), (2, 4), (12, 1)):          # a row of 200 floats = 4 loads per lane: whole steps, half steps, padded steps
        ms, by, ws = ix.replay_roof(tr, slots, kb, rpg, word_sum=True)
        assert ms > 0 and by == float(st[:, 0].sum()) * ix.meta.dim * 4 and ws == want_sum, (kb, rpg, ws, want_sum)
    with pytest.raises(Exception):
        ix.replay_roof(tr, slots, 5, 3)                        # not a shape the replay has: refused
    ix.close()


@pytest.mark.parametrize("func", FUNCS)
def test_wide_beams_equal_the_oracle(func, monkeypatch):
    """Beams of thousands up to the whole index and beyond (device_search_wide.h: the two sets with chunk extremes, candidate
    compaction, bitonic output sort): ids, distance bits, E_q, H_q == the oracle's, vacuumed rows dropped, and a beam wider than the
    index is the beam of the index size.  What the scan's efSearch doubling (embedding.c:329-343) reaches on a large index."""
    import torch
    n, dim, m = 30000, 32, 8
    port, X = build_port(n, dim, m, 32, func, k=60, seed=910 + func)
    for d in (5, 1234, 20000):
        port.set_deleted(d, True)
    Q = gmm(6, dim, k=60, seed=911, stream=1)
    ix = mirror(port, func)
    dQ = torch.from_numpy(Q).cuda()
    for ef in (2100, 9000, 29999, 30000, 100000):
        out = ix.search_torch(dQ, ef, stats=True)
        torch.cuda.synchronize()
        assert "kernel_wide" in ix.last_search_kernel()
        want = port.search_many(Q, ef, nthreads=6)
        cnt = out["counts"].cpu().numpy()
        assert (cnt == want["counts"]).all(), (ef, cnt, want["counts"])
        lab, dst = out["labels"].cpu().numpy().view(np.uint64), out["dists"].cpu().numpy()
        for q in range(6):
            k = cnt[q]
            assert (lab[q, :k] == want["labels"][q, :k]).all() and (bits(dst[q, :k]) == bits(want["dists"][q, :k])).all(), (ef, q)
            assert (lab[q, k:] == pg.NO_LABEL).all()
        st = out["stats"].cpu().numpy().astype(np.uint32)
        assert (st[:, 0] == want["evals"]).all() and (st[:, 1] == want["hops"]).all()
    assert cnt.max() >= n - 500                                 # the last beam returned everything the entry point reaches, minus the vacuumed rows
    ix.close()


@pytest.mark.parametrize("ef", [1, 7, 64, 300])
def test_wide_beam_form_at_small_beams_and_many_queries(ef, monkeypatch):
    """the same form forced on small beams (HNSW_GPU_WIDE_EF_MIN=0) over many queries and equal distances (0/1 vectors: the
    bound is shared by hundreds of elements, so evictions, the candidate compaction and the (dist, label) output order all meet ties)"""
    monkeypatch.setenv("HNSW_GPU_WIDE_EF_MIN", "0")
    rng = np.random.default_rng(12)
    X = rng.integers(0, 2, size=(4000, 7)).astype(np.float32)
    Q = rng.integers(0, 2, size=(300, 7)).astype(np.float32)
    port = oracle.PortIndex(7, 8, 40, 64, pg.DIST_L2)
    port.add(X)
    port.set_deleted(17, True)
    ix = mirror(port, pg.DIST_L2)
    assert_same_as_oracle(ix, port, Q, ef)
    assert "kernel_wide" in ix.last_search_kernel()
    # the walk itself (searchBaseLayer only): pops and element numbers
    gl, gd, gp, ge = ix.search_trace(Q[0], ef, base=True)
    wl, wd, wp, we = port.search_trace(Q[0], ef, base=True)
    assert (gp == wp).all() and ge == we and (gl == wl).all() and (bits(gd) == bits(wd)).all()
    ix.close()


@pytest.mark.skipif(not oracle.have_ref(), reason="needs oracle/_ref (the compiled reference)")
@pytest.mark.parametrize("func,dim", [(pg.DIST_L2, 128), (pg.DIST_L2, 768), (pg.DIST_MANHATTAN, 100), (pg.DIST_COSINE, 100), (pg.DIST_COSINE, 768)])
def test_reference_order_mode_returns_the_compiled_references_id_lists(func, dim, monkeypatch):
    """VERDICT r2 "missing" #6: with HNSW_GPU_REF_ORDER=1 the kernels sum a distance in the order oracle/_ref's own build of
    distfunc.c sums it (8 accumulators, d0^2 + d1^2 per 16 floats, no FMA, its reduction tree; Manhattan: 4 accumulators; cosine: three
    sets of 4, separate multiply and add) — then
    every query's id list AND its distance bits equal the compiled reference's, directly (the default arithmetic gets there
    through the canonical-order oracle and a classification of near-ties).  Debug mode: it reproduces ONE compiler's output."""
    import torch
    n, nq = 20000, 1500
    X = gmm(n, dim, k=100, seed=77 + dim)
    Q = gmm(nq, dim, k=100, seed=78 + dim, stream=1)
    # is this host's _ref build the order the kernel restates?  (a different gcc may vectorise differently)
    sample = oracle.ref_dist_many(func, Q[0], X[:64])
    ref = oracle.RefIndex(dim, 16, 48, 64, func, capacity=n)
    ref.add(X)
    meta = pg.make_meta(dim, 16, 48, 64, func)
    ix = pg.GpuIndex.from_flat(meta, ref.raw(), n)
    monkeypatch.setenv("HNSW_GPU_REF_ORDER", "1")
    d_dev = pg.dist_batch(func, Q[0], X[:64])                   # (canonical order: only to show the two orders do differ somewhere)
    for ef in (64, 128):
        want = ref.search_many(Q, ef, nthreads=8)
        out = ix.search_torch(torch.from_numpy(Q).cuda(), ef)
        torch.cuda.synchronize()
        assert any(f"kernel_beam<{c}" in ix.last_search_kernel() for c in (3, 4, 5))
        lab = out["labels"].cpu().numpy().view(np.uint64)
        dst = out["dists"].cpu().numpy()
        cnt = out["counts"].cpu().numpy()
        if not (bits(dst[0, :cnt[0]]) == bits(oracle.ref_dist_many(func, Q[0], X[lab[0, :cnt[0]].astype(np.int64)]))).all():
            foreign_toolchain("the distance bits of the first query differ between HNSW_GPU_REF_ORDER=1 and oracle/_ref")
        assert (cnt == want["counts"]).all()
        same = (lab == want["labels"]).all(axis=1)
        assert same.all(), f"{int((~same).sum())} of {nq} id lists differ from the compiled reference's"
        for q in range(0, nq, 97):
            assert (bits(dst[q, :cnt[q]]) == bits(oracle.ref_dist_many(func, Q[q], X[lab[q, :cnt[q]].astype(np.int64)]))).all()
    ix.close()


def test_a_launch_reports_the_clock_it_ran_at_and_the_mirror_sits_in_one_aligned_block():
    """include/hnsw_gpu_diag.h (round 6, profiles/r5af_*: a launch state that depended on the process's history): the launch's first
    wave stamps the shader clock and the constant clock when it starts and when it leaves — narrow rows (one wave per query, the
    issue-bound kernel whose time scales with that clock) and wide rows (team form) — and rows | links | labels are three 2 MiB-aligned
    pieces of ONE allocation, before and after the mirror grows; results are unchanged by a reserve."""
    for dim, m, n, nq in ((128, 8, 6000, 4000), (768, 16, 3000, 600)):
        port, X = build_port(n, dim, m, 48, pg.DIST_L2, seed=dim)
        Q = gmm(nq, dim, k=50, seed=dim, stream=1)
        ix = mirror(port, pg.DIST_L2)
        labels, dists, counts = assert_same_as_oracle(ix, port, Q[:64], 64)
        ix.search(Q, 64)
        mhz = ix.last_search_clock_mhz()
        assert 300.0 < mhz < 3500.0, (dim, mhz, ix.last_search_kernel())
        p0 = ix.placement()
        ix.reserve(3 * n)
        p1 = ix.placement()
        for p in (p0, p1):
            assert p["aligned_2MiB"], p
            lo, size = p["arena"]
            assert all(lo <= p[k][0] and p[k][0] + p[k][1] <= lo + size for k in ("rows", "links", "labels")), p
        assert p1["rows"][1] == 3 * p0["rows"][1]
        l2, d2, c2 = ix.search(Q[:64], 64)
        assert (l2 == labels).all() and (bits(d2) == bits(dists)).all() and (c2 == counts).all()
        ix.close()
