"""One scenario on the SIMT-emulated library (tests/emu/hip/hip_runtime.h): the product's own kernel source executed on
host threads, compared with the oracle bit for bit.  Run as a subprocess by tests/test_simt_emu.py (the library is chosen
by environment before pg_embedding_amd is imported).  Prints one JSON line.

    python tests/emu/run_emu_case.py forms|second_walk [emulated-library]
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu                                           # noqa: E402

os.environ["PGEMB_GPU_LIB"] = sys.argv[2] if len(sys.argv) > 2 else build_emu.build()
import numpy as np                                         # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
import util as U                                           # noqa: E402
import oracle                                              # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402

KEYS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_WPB", "HNSW_GPU_MAX_BLOCKS", "HNSW_GPU_NARROW5", "HNSW_GPU_HASH_ENTRIES", "HNSW_GPU_BEAM",      # (HNSW_GPU_TEAM_SPEC: as the caller of this script set it)
        "HNSW_GPU_BEAM16", "HNSW_GPU_FORCE_LDS_HEAPS", "SIMT_EMU_CUS", "SIMT_EMU_JITTER", "SIMT_EMU_JITTER_US", "SIMT_EMU_SEED", "HNSW_GPU_INSERT_FUSED")


def setenv(env):
    for k in KEYS:
        os.environ.pop(k, None)
    os.environ.update(env)


def wrong(got, want, nq, stats=None):
    l2, d2, c2 = got
    bad = 0
    for q in range(nq):
        same = (l2[q] == want["labels"][q]).all() and (U.bits(d2[q]) == U.bits(want["dists"][q])).all() and c2[q] == want["counts"][q]
        bad += 0 if same else 1
    return bad


def forms():
    """every kernel form the host can pick, three metrics, odd and even row widths: ids, distance bits and counts == oracle"""
    out = []
    variants = [{}, {"HNSW_GPU_TEAM": "0"}, {"HNSW_GPU_TEAM": "0", "HNSW_GPU_NARROW5": "0"}, {"HNSW_GPU_TEAM": "1"},
                {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "4"}, {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "2", "HNSW_GPU_HASH_ENTRIES": "512"},
                {"HNSW_GPU_HASH_ENTRIES": "0"}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}, {"HNSW_GPU_TEAM": "1", "SIMT_EMU_CUS": "64"}]
    cfgs = ((32, 8, pg.DIST_L2, (10, 100)), (100, 16, pg.DIST_COSINE, (40,)), (200, 8, pg.DIST_MANHATTAN, (40, 300)), (768, 16, pg.DIST_L2, (64,)))
    if os.environ.get("EMU_FORMS_QUICK"):
        cfgs = (cfgs[1], cfgs[3])
    for dim, m, func, efs in cfgs:
        n, nq = (1200, 12) if dim < 700 else (500, 6)
        port, X = U.build_port(n, dim, m, 40, func, k=10, seed=dim)
        Q = gmm(nq, dim, k=10, seed=dim + 1)
        for k, ef in enumerate(efs):
            ix = U.mirror(port, func, efs=ef)
            want = port.search_many(Q, ef, nthreads=4)
            for env in (variants if k == 0 else variants[::3]):      # (a configuration's second beam width: every third variant)
                setenv(env)
                t0 = time.time()
                got = ix.search(Q, ef)
                out.append({"dim": dim, "func": int(func), "ef": ef, "env": env, "kernel": ix.last_search_kernel(), "wrong": wrong(got, want, nq),
                            "seconds": round(time.time() - t0, 2)})
            ix.close()
    return out


def accept():
    """the accept decisions of a hop depend on each other when the beam is small or distances tie: beams of 1-9 over 32-link lists (every hop
    brings rows that push each other out), once on quantised rows (exact ties: the `<=` of the count), every beam-kernel form; ids, distance
    bits, counts == oracle, and the walk's pop sequence + evaluation count for the first queries (hnsw_gpu_search_trace)"""
    out = []
    variants = [{}, {"HNSW_GPU_TEAM": "0"}, {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "4"}, {"HNSW_GPU_TEAM": "0", "HNSW_GPU_NARROW5": "0"}]
    cfgs = ((24, 16, pg.DIST_L2, False), (24, 16, pg.DIST_L2, True), (40, 16, pg.DIST_COSINE, False), (200, 12, pg.DIST_MANHATTAN, True))
    quick = bool(os.environ.get("EMU_ACCEPT_QUICK"))               # (the teeth runs: the configuration in which a wrong decision shows most often)
    if quick:
        cfgs, variants = cfgs[1:2], variants[:2]
    for dim, m, func, quant in cfgs:
        n, nq = 1500, 24
        X = gmm(n, dim, k=6, seed=900 + dim)
        Q = gmm(nq, dim, k=6, seed=901 + dim)
        if quant:
            X, Q = np.round(2.0 * X).astype(np.float32), np.round(2.0 * Q).astype(np.float32)
        port = oracle.PortIndex(dim, m, 60, 16, func)
        port.add(X, np.arange(n, dtype=np.uint64) + 7)
        for ef in ((1, 2, 5) if quick else (1, 2, 3, 5, 9, 130)):
            ix = U.mirror(port, func, efs=ef)
            want = port.search_many(Q, ef, nthreads=4)
            for env in variants:
                setenv(env)
                t0 = time.time()
                got = ix.search(Q, ef)
                bad = wrong(got, want, nq)
                tbad = 0
                for q in range(4):
                    lab, dst, pops, nev = ix.search_trace(Q[q], ef)
                    tbad += 0 if (len(pops) == want["hops"][q] and nev == want["evals"][q]) else 1
                out.append({"dim": dim, "func": int(func), "quantised": quant, "ef": ef, "env": env, "kernel": ix.last_search_kernel(), "wrong": bad,
                            "trace_wrong": tbad, "seconds": round(time.time() - t0, 2)})
            ix.close()
    return out


def second_walk():
    """ONE wave walks 32 neighbouring queries one after the other with seven helpers attached (a launch squeezed into one
    block on an emulated 256-CU device): a helper that slept through the gap between two walks must not feed the next one."""
    n, dim, m, ef = 3000, 96, 16, 48
    port, X = U.build_port(n, dim, m, 40, pg.DIST_L2, k=10, seed=3)
    os.environ["SIMT_EMU_CUS"] = "256"                     # (a mirror remembers its device's CU count)
    ix = U.mirror(port, pg.DIST_L2, efs=ef)
    rng = np.random.default_rng(5)
    base = X[rng.integers(0, n, size=4)]
    Q = (np.repeat(base, 8, axis=0) + 0.01 * rng.standard_normal((32, dim))).astype(np.float32)
    want = port.search_many(Q, ef, nthreads=4)
    out = []
    envs = ({"HNSW_GPU_TEAM": "1", "HNSW_GPU_MAX_BLOCKS": "1", "SIMT_EMU_CUS": "256"},
            {"HNSW_GPU_TEAM": "1", "HNSW_GPU_MAX_BLOCKS": "1", "SIMT_EMU_CUS": "256", "SIMT_EMU_JITTER": "500", "SIMT_EMU_JITTER_US": "3000"},
            {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": "4", "HNSW_GPU_MAX_BLOCKS": "2", "SIMT_EMU_CUS": "256"})
    quick = bool(os.environ.get("EMU_SECOND_WALK_QUICK"))      # (the report-only and teeth variants: one schedule with, one without jitter, once)
    for env in (envs[:2] if quick else envs):
        bad = 0
        for rep in range(1 if quick else 2):
            setenv(dict(env, SIMT_EMU_SEED=str(rep)))
            bad += wrong(ix.search(Q, ef), want, 32)
        out.append({"env": env, "kernel": ix.last_search_kernel(), "walks": 32 if quick else 64, "wrong": bad, "blocks_x_waves": ix.last_search_slots(), "health": ix.health()})
    return out


def moving_helpers():
    """several walking waves per block whose helpers change walks (a wave whose walk is over helps a sibling that still walks),
    few helpers per walk and all of them slice helpers (HNSW_GPU_TEAM_SPEC 0 / 2): a helper's completion word must not be taken
    for another walking wave's job with the same number"""
    n, dim, m, ef = 3000, 96, 16, 48
    port, X = U.build_port(n, dim, m, 40, pg.DIST_L2, k=10, seed=3)
    out = []
    for nq in (6, 10, 14):
        Q = gmm(nq, dim, k=10, seed=6 + nq)
        want = port.search_many(Q, ef, nthreads=4)
        for spec in ("0", "2"):
            setenv({"HNSW_GPU_TEAM": "1", "SIMT_EMU_CUS": "2"})
            os.environ["HNSW_GPU_TEAM_SPEC"] = spec
            ix = U.mirror(port, pg.DIST_L2, efs=ef)
            bad = 0
            for rep in range(3):
                bad += wrong(ix.search(Q, ef), want, nq)
            out.append({"nq": nq, "spec": spec, "wrong": bad, "walks": 3 * nq, "health": ix.health(), "kernel": ix.last_search_kernel()})
            ix.close()
    os.environ.pop("HNSW_GPU_TEAM_SPEC", None)
    return out


def wide():
    """the wide-beam form (device_search_wide.h) forced on every beam (HNSW_GPU_WIDE_EF_MIN=0): small beams with ties, beams of
    hundreds, the index size and beyond, vacuumed rows, the walk's pop sequence — all == the oracle"""
    import oracle
    out = []
    setenv({})
    os.environ["HNSW_GPU_WIDE_EF_MIN"] = "0"
    for dim, m, func, n, efs in ((24, 8, pg.DIST_L2, 900, (1, 40, 899, 5000)), (40, 6, pg.DIST_COSINE, 600, (64, 700)), (6, 8, pg.DIST_L2, 1500, (64, 256))):
        if dim == 6:
            rng = np.random.default_rng(11)
            X = rng.integers(0, 2, size=(n, dim)).astype(np.float32)          # ties: 7 distinct distances
            port = oracle.PortIndex(dim, m, 40, 64, func)
            port.add(X)
            Q = rng.integers(0, 2, size=(8, dim)).astype(np.float32)
        else:
            port, X = U.build_port(n, dim, m, 40, func, k=10, seed=dim)
            Q = gmm(8, dim, k=10, seed=dim + 1)
        for d in (3, 77, 200):
            port.set_deleted(d, True)
        for ef in efs:
            ix = U.mirror(port, func, efs=min(ef, 64))
            want = port.search_many(Q, ef, nthreads=4)
            l, d, c = ix.search(Q, ef)
            bad = 0
            for q in range(len(Q)):
                k = want["counts"][q]
                ok = c[q] == k and (l[q][:k] == want["labels"][q][:k]).all() and (U.bits(d[q][:k]) == U.bits(want["dists"][q][:k])).all()
                bad += 0 if ok else 1
            gl, gd, gp, ge = ix.search_trace(Q[0], ef)
            wl, wd, wp, we = port.search_trace(Q[0], ef)
            tr = int(not (len(gp) == len(wp) and (gp == wp).all() and ge == we))
            out.append({"dim": dim, "func": int(func), "ef": ef, "kernel": ix.last_search_kernel(), "wrong": bad, "trace_wrong": tr})
            ix.close()
    os.environ.pop("HNSW_GPU_WIDE_EF_MIN", None)
    return out


def reforder():
    """debug arithmetic HNSW_GPU_REF_ORDER=1 (device_dist.h, score_rows_ref): the summation order of oracle/_ref's own build of
    distfunc.c.  With it the kernels' id lists AND distance bits equal the compiled reference's for every query — not through
    the canonical-order oracle, directly.  (Skipped when this host's _ref build sums in another order: checked first on
    plain pairs against a numpy statement of the order.)"""
    import oracle
    if not oracle.have_ref():
        return {"skipped": "no oracle/_ref here"}
    out = []
    # (round 6: the production load shape + transposed accumulation — partial slices (48, 36, 44 dims), whole slices (128), a whole load
    # batch of the widest shape (768) and two of them (1536), every function in a narrow and in a wide shape)
    for func, dim, n in ((pg.DIST_L2, 128, 2500), (pg.DIST_L2, 48, 1500), (pg.DIST_MANHATTAN, 36, 1500), (pg.DIST_COSINE, 44, 1500),
                         (pg.DIST_COSINE, 768, 500), (pg.DIST_L2, 1536, 400), (pg.DIST_MANHATTAN, 300, 600), (pg.DIST_L2, 272, 600)):
        X = gmm(n, dim, k=20, seed=dim)
        Q = gmm(24, dim, k=20, seed=dim + 1)
        ref = oracle.RefIndex(dim, 8, 32, 64, func, capacity=n)
        ref.add(X)
        # the order itself, in numpy float32, against the reference's own hnsw_dist_func on this host
        def np_dist(q, x):
            q = q.astype(np.float32); x = x.astype(np.float32)
            if func == pg.DIST_L2:
                acc = np.zeros(8, np.float32)
                for k in range(0, dim, 16):
                    d0, d1 = q[k:k + 8] - x[k:k + 8], q[k + 8:k + 16] - x[k + 8:k + 16]
                    acc = acc + (d0 * d0 + d1 * d1)
                r = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[6] + acc[7]) + (acc[4] + acc[5]))
                return np.sqrt(np.float32(r))
            if func == pg.DIST_COSINE:
                dot, na, nb = np.zeros(4, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32)
                for k in range(0, dim, 4):
                    dot = dot + q[k:k + 4] * x[k:k + 4]
                    na = na + q[k:k + 4] * q[k:k + 4]
                    nb = nb + x[k:k + 4] * x[k:k + 4]
                red = lambda a: (a[0] + a[2]) + (a[1] + a[3])
                return np.float32(1.0 - np.float64(red(dot)) / np.sqrt(np.float64(np.float32(red(na) * red(nb)))))
            acc = np.zeros(4, np.float32)
            for k in range(0, dim, 4):
                acc = acc + np.abs(q[k:k + 4] - x[k:k + 4])
            return (acc[0] + acc[2]) + (acc[1] + acc[3])
        same_order = all(U.bits(np.float32(oracle.ref_dist(func, Q[i % 24], X[i * 7 % n]))) == U.bits(np.float32(np_dist(Q[i % 24], X[i * 7 % n]))) for i in range(200))
        if not same_order:
            out.append({"func": int(func), "dim": dim, "skipped": "this host's _ref build sums in another order"})
            continue
        setenv({})
        os.environ["HNSW_GPU_REF_ORDER"] = "1"
        meta = pg.make_meta(dim, 8, 32, 64, func)
        ix = pg.GpuIndex.from_flat(meta, ref.raw(), n)
        bad = 0
        for ef in (10, 64, 128):
            want = ref.search_many(Q, ef)
            l, d, c = ix.search(Q, ef)
            for q in range(len(Q)):
                k = want["counts"][q]
                ok = c[q] == k and (l[q][:k] == want["labels"][q][:k]).all()
                # (hnsw_search returns no distances: the reference's own hnsw_dist_func for the rows it returned)
                ok = ok and (U.bits(d[q][:k]) == U.bits(oracle.ref_dist_many(func, Q[q], X[want["labels"][q][:k].astype(np.int64)]))).all()
                bad += 0 if ok else 1
        out.append({"func": int(func), "dim": dim, "kernel": ix.last_search_kernel(), "wrong_vs_the_compiled_reference": bad, "queries": 3 * len(Q)})
        ix.close()
        os.environ.pop("HNSW_GPU_REF_ORDER", None)
    return out


def abort():
    """the host's abort word: a launch that is asked to end does end (every wave leaves at its next look), says so in the
    health words, and the next launch on the same workspace is exact again (the bitmaps the aborted waves left are re-zeroed)"""
    import threading
    n, dim, m, ef = 3000, 96, 16, 48
    port, X = U.build_port(n, dim, m, 40, pg.DIST_L2, k=10, seed=3)
    out = []
    for env in ({"HNSW_GPU_TEAM": "1"}, {"HNSW_GPU_TEAM": "0", "HNSW_GPU_HASH_ENTRIES": "0"}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}):
        setenv(dict(env, SIMT_EMU_CUS="2"))
        ix = U.mirror(port, pg.DIST_L2, efs=ef)
        Q = gmm(4000, dim, k=10, seed=9)                       # minutes of emulated work if nobody stops it
        t = threading.Timer(1.0, lambda: ix.abort())
        t0 = time.time()
        t.start()
        said = ""
        try:
            ix.search(Q, ef)
        except RuntimeError as e:                              # the host-pointer form reports an interrupted launch (include/hnsw_gpu.h)
            said = str(e)
        took = time.time() - t0
        h = ix.health()
        Q2 = gmm(12, dim, k=10, seed=10)
        want = port.search_many(Q2, ef, nthreads=4)
        bad = wrong(ix.search(Q2, ef), want, 12)
        out.append({"env": env, "kernel": ix.last_search_kernel(), "seconds_until_the_launch_ended": round(took, 2), "health_after_abort": h,
                    "health_after_next": ix.health(), "wrong_after": bad, "error_of_the_interrupted_call": said})
        ix.close()
    return out


def traced():
    """the evaluation trace of a launch (measurement entry point of bench.py's replay roof) == the rows the walk must score,
    restated from its pop sequence and the link lists; and the replay kernel runs over it"""
    import ctypes as C
    out = []
    n, dim, m, ef, cap = 1500, 40, 8, 48, 512
    port, X = U.build_port(n, dim, m, 40, pg.DIST_L2, k=10, seed=13)
    Q = gmm(6, dim, k=10, seed=14)
    for env in ({"HNSW_GPU_TEAM": "0"}, {"HNSW_GPU_TEAM": "1"}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}):
        setenv(env)
        ix = U.mirror(port, pg.DIST_L2, efs=ef)
        nq = len(Q)
        lab = np.empty((nq, ef), np.uint64); dst = np.empty((nq, ef), np.float32); cnt = np.empty(nq, np.uint32)
        st = np.zeros((nq, 2), np.uint32); ev = np.full((nq, cap), 0xFFFFFFFF, np.uint32); tm = np.zeros((nq, 2), np.uint64)
        Qc = np.ascontiguousarray(Q, np.float32)
        rc = ix.L.hnsw_gpu_search_traced_dev(ix._h, Qc.ctypes.data, nq, ef, lab.ctypes.data, dst.ctypes.data, cnt.ctypes.data, st.ctypes.data,
                                             ev.ctypes.data, cap, tm.ctypes.data, None)       # (emulated device memory IS host memory)
        assert rc == 0, rc
        bad = 0
        for i in range(nq):
            _, _, pops, nev = ix.search_trace(Q[i], ef)
            want = U.evals_from_pops(port.raw(), ix.meta, n, ix.meta.enterpoint_node, pops)
            bad += 0 if (st[i, 0] == len(want) == nev and (ev[i, :len(want)] == want).all() and tm[i, 1] >= tm[i, 0] > 0) else 1
        ms, by, ws = C.c_float(0), C.c_double(0), C.c_uint64(0)
        rc = ix.L.hnsw_gpu_replay_roof(ix._h, ev.ctypes.data, cap, st.ctypes.data, nq, 8, 2, 4, C.byref(ms), C.byref(by), C.byref(ws))
        words = X.view(np.uint32).astype(np.uint64).sum(axis=1)
        want_sum = int(sum(int(words[ev[i, :st[i, 0]]].sum()) for i in range(nq)) % (1 << 64))
        # the same trace cut into 2 / 4 pieces per query, each gathered by a wave of its own (hnsw_gpu_replay_roof_parts: the roof of a
        # launch with fewer queries than resident waves): the same bytes, the same words
        parts_ok = True
        for parts in (2, 4):
            ms2, by2, ws2 = C.c_float(0), C.c_double(0), C.c_uint64(0)
            rc2 = ix.L.hnsw_gpu_replay_roof_parts(ix._h, ev.ctypes.data, cap, st.ctypes.data, nq, 8, 2, 4, parts, C.byref(ms2), C.byref(by2), C.byref(ws2))
            parts_ok = parts_ok and rc2 == 0 and by2.value == by.value and int(ws2.value) == want_sum
        # walking waves per block of a small team launch (hnsw_gpu_ctx_set_walkers, what the server's lanes say): nothing but the shape changes
        walkers_ok = True
        if env.get("HNSW_GPU_TEAM") == "1":
            ctx = pg.SearchContext(ix)
            wantq = port.search_many(Q, ef, nthreads=2)
            for w in (0, 1, 3, 8):
                ctx.set_walkers(w)
                walkers_ok = walkers_ok and wrong(ctx.search_host(Q, ef), wantq, nq) == 0
            ctx.close()
        out.append({"env": env, "kernel": ix.last_search_kernel(), "wrong": bad, "replay_rc": rc, "replay_bytes": by.value,
                    "want_bytes": float(st[:, 0].sum()) * dim * 4, "word_sum_ok": int(ws.value) == want_sum, "parts_ok": parts_ok,
                    "walkers_ok": walkers_ok})
        ix.close()
    return out


def sharded():
    """hnsw_gpu_sharded_create / _search[_dev] with the shards on SEVERAL emulated devices (SIMT_EMU_DEVICES, set by the caller
    before the library loads): per-shard search on the shard's own device and stream, results into the merging device's buffer
    through peer access or — SIMT_EMU_PEER=0 / HNSW_GPU_SHARDED_NO_PEER=1 — through a staged peer copy, one merge == oracle per
    shard + CPU merge.  The emulator ends the process if a kernel touches another device's memory without peer access, and
    fails a launch or event record on a stream of the wrong device."""
    import ctypes as C
    from pg_embedding_amd._lib import gpu_lib
    ndev = gpu_lib().hnsw_gpu_device_count()
    out = {"devices": ndev, "cases": []}
    n, dim, m, efc, ef, nq = 1800, 24, 6, 32, 20, 24
    X = gmm(n, dim, k=12, seed=41)
    Q = gmm(nq, dim, k=12, seed=41, stream=1)
    X[n - 5] = X[7]                                            # identical rows in different shards
    meta = pg.make_meta(dim, m, efc, ef, pg.DIST_L2)
    import oracle
    for nshards, order in ((2, (0, 1)), (3, (0, 1, 2)), (3, (2, 0, 1)), (5, (1, 0, 2, 1, 0))):
        if max(order) >= ndev:
            continue
        shards, per = [], []
        for r in range(nshards):
            lo, hi = n * r // nshards, n * (r + 1) // nshards
            port = oracle.PortIndex(dim, m, efc, ef, pg.DIST_L2)
            port.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))
            port.set_deleted(1)
            per.append(port.search_many(Q, ef))
            shards.append(pg.GpuIndex.from_flat(meta, port.raw(), hi - lo, device=order[r]))
        sh = pg.LocalShardedIndex(shards)
        bad = 0
        for rep in range(2):                                   # second call: buffers and events are reused
            ml, md, mc = sh.search(Q, ef)
            for q in range(nq):
                l = np.concatenate([p["labels"][q, :p["counts"][q]] for p in per])
                d = np.concatenate([p["dists"][q, :p["counts"][q]] for p in per])
                o = np.lexsort((l, d))[:ef]
                ok = mc[q] == o.size and (ml[q, :o.size] == l[o]).all() and (U.bits(md[q, :o.size]) == U.bits(d[o])).all()
                bad += 0 if ok else 1
        out["cases"].append({"shards": nshards, "devices_of_shards": list(order), "wrong": bad})
        sh.close()
        for ix in shards:
            ix.close()
    return out


def others():
    """the other kernels behind the C-ABI: serial device insert (graph bytes == the oracle's), batched insert (searchable),
    the walk's pop sequence, vacuum flags, export, the canonical exhaustive scan"""
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_build import live_image
    out = {}
    setenv({})
    for func, dim, m, efc, n in ((pg.DIST_L2, 12, 4, 16, 180), (pg.DIST_COSINE, 9, 1, 5, 120), (pg.DIST_MANHATTAN, 20, 3, 40, 140)):
        X = gmm(n, dim, k=10, seed=dim)
        labels = np.arange(n, dtype=np.uint64) * 7 + 5
        port = oracle.PortIndex(dim, m, efc, 64, func)
        port.add(X, labels)
        meta = pg.make_meta(dim, m, efc, 64, func)
        ix = pg.GpuIndex.empty(meta, n)
        ix.append(X[:100], labels[:100])
        ix.link(0, 100, max_batch=1)
        ix.append(X[100:], labels[100:])
        ix.link(100, n - 100, max_batch=1)
        got = ix.export_flat().reshape(n, -1)
        want = live_image(port.raw(), meta, n)
        out[f"serial_insert_{func}_{dim}"] = int((got != want).any(axis=1).sum())
        # the walk itself: pops and evaluation count
        q = gmm(1, dim, k=10, seed=99)[0]
        gl, gd, gp, ge = ix.search_trace(q, 24)
        wl, wd, wp, we = port.search_trace(q, 24)
        out[f"trace_{func}_{dim}"] = int(not (len(gp) == len(wp) and (gp == wp).all() and ge == we and (gl == wl).all() and (U.bits(gd) == U.bits(wd)).all()))
        # vacuum flags: flagged elements leave the results, the walk stays
        dead = [int(x) for x in gl[:3] // 7]
        ix.set_deleted_many(np.asarray(dead, np.uint32), True)
        for d in dead:
            port.set_deleted(d, True)
        l2, d2, c2 = ix.search(q.reshape(1, -1), 24)
        w2 = port.search_many(q.reshape(1, -1), 24)
        c = int(c2[0])
        out[f"vacuum_{func}_{dim}"] = int(not (c == w2["counts"][0] and c < 24 and (l2[0][:c] == w2["labels"][0][:c]).all()))
        ix.close()
    # batched insert: not the reference's order, but a graph the search finds its way in (recall against the exact scan)
    n, dim = 800, 24
    X = gmm(n, dim, k=10, seed=4)
    ix = pg.GpuIndex.empty(pg.make_meta(dim, 8, 40, 40, pg.DIST_L2), n)
    ix.append(X)
    ix.link(0, n)
    Q = gmm(16, dim, k=10, seed=5)
    lab, dist, cnt = ix.search(Q, 40)
    exact = np.argsort(((Q[:, None, :] - X[None, :, :]) ** 2).sum(-1), axis=1)[:, :10]
    out["batched_insert_recall_at_10"] = float(np.mean([len(set(map(int, lab[i][:10])) & set(map(int, exact[i]))) / 10 for i in range(16)]))
    # diagnostics of include/hnsw_gpu_diag.h: the launch's first wave stamped both clocks (the emulator's two clocks are 32 : 1), and the
    # mirror's three arrays sit inside ONE allocation on 2 MiB boundaries, also after the mirror has grown
    pl = ix.placement()
    ix.reserve(3 * n)
    pl2 = ix.placement()
    lab2, dist2, cnt2 = ix.search(Q, 40)
    inside = lambda p: all(p["arena"][0] <= p[k][0] and p[k][0] + p[k][1] <= p["arena"][0] + p["arena"][1] for k in ("rows", "links", "labels"))
    out["diag_clock_and_placement"] = int(not (ix.last_search_clock_mhz() > 0.0 and pl["aligned_2MiB"] and pl2["aligned_2MiB"] and inside(pl) and inside(pl2)
                                               and pl2["rows"][1] == 3 * pl["rows"][1] and (lab2 == lab).all() and (U.bits(dist2) == U.bits(dist)).all()))
    ix.close()
    return out


def insert():
    """the serial insert in one call (device_insert.h: pair triangle + chain, one block per target): hnsw_gpu_index_insert_one and
    hnsw_gpu_index_insert_candidates row by row == the oracle's graph bytes, and the lists they return == the mirror's; the four-launch
    builder path (HNSW_GPU_INSERT_FUSED=0) gives the same bytes"""
    import ctypes as C
    import oracle
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_gpu_build import live_image
    out = {}
    quick = os.environ.get("EMU_INSERT_QUICK") == "1"
    shake = {k: os.environ[k] for k in ("SIMT_EMU_JITTER", "SIMT_EMU_JITTER_US", "SIMT_EMU_SEED") if k in os.environ}   # the caller's schedule shaking stays on
    cases = ((pg.DIST_L2, 12, 4, 36, 150), (pg.DIST_COSINE, 9, 1, 5, 90), (pg.DIST_MANHATTAN, 20, 3, 40, 110),
             (pg.DIST_L2, 6, 40, 100, 130),            # lists longer than a wavefront (maxM = 80), two mask words per candidate
             (pg.DIST_L2, 6, 4, 600, 100))             # more candidates than the chain keeps (512): the general builder path, same bytes
    for func, dim, m, efc, n in (cases[:1] if quick else cases):
        X = gmm(n, dim, k=10, seed=dim + 1)
        X[40:44] = X[20:24]                                  # equal distances: ties by element number
        labels = np.arange(n, dtype=np.uint64) * 5 + 2
        port = oracle.PortIndex(dim, m, efc, 64, func)
        port.add(X, labels)
        meta = pg.make_meta(dim, m, efc, 64, func)
        maxM = int(meta.maxM)
        want = live_image(port.raw(), meta, n)
        for mode, fused in (("one", "1"), ("candidates", "1"), ("one", "0")):
            if quick and mode == "one" and fused == "0":
                continue
            setenv(dict(shake, HNSW_GPU_INSERT_FUSED=fused))
            ix = pg.GpuIndex.empty(meta, n)
            mine = (C.c_uint32 * (maxM + 1))()
            others = (C.c_uint32 * (maxM * (maxM + 1)))()
            bad_lists = 0
            for i in range(n):
                p = np.ascontiguousarray(X[i])
                if mode == "candidates" and i > 0:
                    ci, cd, pops, nev = ix.search_trace(p, efc, base=True)
                    ci32 = np.ascontiguousarray(ci.astype(np.uint32))
                    cd32 = np.ascontiguousarray(cd, dtype=np.float32)
                    rc = ix.L.hnsw_gpu_index_insert_candidates(ix._h, p.ctypes.data, int(labels[i]), i, ci32.ctypes.data, cd32.ctypes.data,
                                                               len(ci32), mine, others)
                else:
                    rc = ix.L.hnsw_gpu_index_insert_one(ix._h, p.ctypes.data, int(labels[i]), i, mine, others)
                if rc != 0:
                    out[f"rc_f{func}_efc{efc}_{mode}_{fused}"] = [i, rc]
                    break
                if i in (1, n // 2, n - 1):
                    got = ix.export_flat().reshape(i + 1, -1)[:, :(maxM + 1) * 4].copy().view(np.uint32)
                    bad_lists += int(not (np.frombuffer(mine, np.uint32)[:1 + mine[0]] == got[i, :1 + mine[0]]).all())
                    for j in range(mine[0]):
                        o = np.frombuffer(others, np.uint32)[j * (maxM + 1):(j + 1) * (maxM + 1)]
                        bad_lists += int(not (o[:1 + o[0]] == got[mine[1 + j], :1 + o[0]]).all())
            got = ix.export_flat().reshape(n, -1)
            out[f"insert_f{func}_d{dim}_m{m}_efc{efc}_{mode}_fused{fused}"] = int((got != want).any(axis=1).sum()) + bad_lists
            ix.close()
    setenv({})
    paths = (C.c_uint64 * 2)()
    pg._lib.gpu_lib().hnsw_gpu_insert_path_counts(paths)
    out["two_launch_inserts"], out["general_inserts"] = int(paths[0]), int(paths[1])
    return out


def part(want, lo, hi):
    return {k: want[k][lo:hi] for k in ("labels", "dists", "counts")}


def stream():
    """streams (include/hnsw_gpu.h): ONE resident launch fed from the host while it runs — the emulator runs that launch on threads of
    its own (doorbell block next to the walking block).  Queries published a few at a time while others are still walking, a ring far
    smaller than the number of queries, both completion forms, one and several walking waves per block, the stream closed and another
    opened, ordinary launches on the mirror before, between and after: every answer == the oracle's"""
    out = []
    for dim, m, func, ef, walkers in ((32, 8, pg.DIST_L2, 24, 2), (100, 12, pg.DIST_COSINE, 40, 8), (200, 8, pg.DIST_MANHATTAN, 16, 1)):
        n, nq, ring = 1200, int(os.environ.get("EMU_STREAM_QUERIES", "150")), 64
        port, X = U.build_port(n, dim, m, 40, func, k=10, seed=dim)
        Q = gmm(nq, dim, k=10, seed=dim + 3)
        want = port.search_many(Q, ef, nthreads=4)
        setenv({"SIMT_EMU_CUS": "2"})
        ix = U.mirror(port, func, efs=ef)
        bad_launch = wrong(ix.search(Q[:8], ef), part(want, 0, 8), 8)
        ctx = pg.SearchContext(ix)
        bad = 0
        bad_by_round = [0, 0]
        for round_ in range(2):
            pg.config_set("HNSW_GPU_STREAM_LIGHT", None if round_ == 0 else 0)
            st = pg.SearchStream(ctx, ef, ring=ring, walkers=walkers)
            try:
                alive_at_start = st.alive()
                rng = np.random.default_rng(round_)
                done, pending = 0, []
                while done < nq or pending:
                    inflight = sum(len(sl) for _, sl in pending)
                    if done < nq and inflight <= ring // 2:
                        k = int(min(nq - done, rng.integers(1, ring // 2 - 1), ring - inflight - 1))
                        pending.append((done, st.submit(Q[done:done + k])))
                        done += k
                        continue
                    first, slots = pending.pop(0)
                    lab, dst, cnt = st.wait(slots, timeout=300.0)
                    k = len(slots)
                    w = wrong((lab, dst, cnt), part(want, first, first + k), k)
                    bad += w
                    bad_by_round[round_] += w
            finally:
                pg.config_set("HNSW_GPU_STREAM_LIGHT", None)
                st.close()
            bad_launch += wrong(ix.search(Q[8:16], ef), part(want, 8, 16), 8)     # the mirror's ordinary launches go on
        out.append({"dim": dim, "func": int(func), "ef": ef, "walkers": walkers, "queries_through_streams": 2 * nq, "wrong": bad, "wrong_by_completion_form": bad_by_round,
                    "wrong_in_ordinary_launches": bad_launch, "alive_while_open": bool(alive_at_start), "health": ix.health()})
        ctx.close()
        ix.close()
    return out


if __name__ == "__main__":
    print(json.dumps({"accept": accept, "forms": forms, "second_walk": second_walk, "others": others, "abort": abort, "traced": traced, "sharded": sharded, "moving_helpers": moving_helpers, "wide": wide, "reforder": reforder, "insert": insert, "stream": stream}[sys.argv[1]]()))
