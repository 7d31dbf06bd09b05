"""Team form under the schedules the ordinary suites never produce, on the device, every answer compared bit
for bit with the port oracle:

  * a wave that walks SEVERAL queries while its siblings help — a launch squeezed into a few blocks (HNSW_GPU_MAX_BLOCKS), so
    that every wave with queries takes many of them through the ticket counter with helpers attached; neighbouring queries of one
    cluster follow each other, so that a package or a slice scored against the previous query would be for elements the next walk
    pops too (the helper-bit clear at the start of a walk, the slice-job protocol);
  * the batching server's shape: one big launch on stream A, back-to-back launches of 1-64 queries on streams B and C through
    search contexts — the small launches' blocks start late and one after the other, behind the big one;
  * the host's abort word on a real launch.

Each test is bounded to about a minute and runs under the suite's watchdogs (tests/conftest.py): every inter-wave wait in the
kernels is bounded, so a protocol mistake shows up as a wrong answer or a health counter here, not as a hang."""
import os
import time

import numpy as np
import pytest

import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from util import bits, build_port, mirror

pytestmark = pytest.mark.gpu

TEAM_KEYS = ("HNSW_GPU_TEAM", "HNSW_GPU_TEAM_WPB", "HNSW_GPU_MAX_BLOCKS", "HNSW_GPU_TEAM_SPEC")


def _setenv(monkeypatch, env):
    for k in TEAM_KEYS:
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)


def _same(out, want, nq, what):
    lab = out["labels"].cpu().numpy().view(np.uint64)
    assert (lab == want["labels"][:nq]).all(), ("labels", what, int((lab != want["labels"][:nq]).any(axis=1).sum()))
    assert (bits(out["dists"].cpu().numpy()) == bits(want["dists"][:nq])).all(), ("distance bits", what)
    assert (out["counts"].cpu().numpy() == want["counts"][:nq]).all(), ("counts", what)
    if out.get("stats") is not None:
        st = out["stats"].cpu().numpy().astype(np.uint32)
        assert (st[:, 0] == want["evals"][:nq]).all() and (st[:, 1] == want["hops"][:nq]).all(), ("E_q / H_q", what)


@pytest.mark.timeout(240, method="thread")
@pytest.mark.parametrize("dim,m,func,n", [(768, 16, pg.DIST_L2, 10000), (96, 16, pg.DIST_L2, 12000), (768, 32, pg.DIST_COSINE, 6000)])
def test_a_wave_that_walks_many_queries_with_helpers_attached(dim, m, func, n, monkeypatch):
    import torch
    rng = np.random.default_rng(11 + dim + func)
    port, X = build_port(n, dim, m, 64, func, k=40, seed=900 + dim + func)
    ix = mirror(port, func, efs=128)
    t_end = time.time() + 20.0                                 # per configuration; the oracle's share is outside the budget
    cases = 0
    delivered = 0
    for r in range(200):
        if time.time() > t_end and r >= 6:
            break
        # queries = slightly moved copies of a few rows, one after the other: consecutive walks cross the same elements
        nq = int(rng.choice([8, 24, 64, 200]))
        base = X[rng.integers(0, n, size=max(1, nq // 8))]
        Q = (np.repeat(base, 8, axis=0)[:nq] + 0.01 * rng.standard_normal((nq, dim))).astype(np.float32)
        ef = int(rng.choice([40, 128]))
        want = port.search_many(Q, ef, nthreads=8)
        dQ = torch.from_numpy(Q).cuda()
        for blocks, wpb, spec in (("1", "8", "5"), ("2", "8", "0"), ("3", "4", "2"), ("1", "2", "0"), ("2", "8", "8")):
            _setenv(monkeypatch, {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_WPB": wpb, "HNSW_GPU_MAX_BLOCKS": blocks, "HNSW_GPU_TEAM_SPEC": spec})
            out = ix.search_torch(dQ, ef, stats=True)
            torch.cuda.synchronize()
            assert ", true, " in ix.last_search_kernel()          # (the TEAM template argument)
            _same(out, want, nq, (dim, r, nq, ef, blocks, wpb, spec))
            cases += 1
    h = ix.health()
    delivered = h["slices_delivered"]
    print(f"\n[team stress {dim}d func {func}] {cases} launches exact; health {h}")
    assert h["aborted_waves"] == 0 and h["abort_pending"] == 0, h
    assert delivered > 0, h                                    # the slice mechanism did run
    assert h["slice_timeouts"] == 0, h                         # ... and no helper was ever late on an otherwise idle device
    ix.close()


@pytest.mark.timeout(300, method="thread")
@pytest.mark.parametrize("dim,m,func,n", [(1536, 32, pg.DIST_COSINE, 5000), (768, 16, pg.DIST_L2, 10000)])
def test_helpers_that_move_from_walk_to_walk_inside_a_block(dim, m, func, n, monkeypatch):
    """Launches of a few hundred to a thousand queries: every block has several walking waves, and a wave whose walk is over helps
    whichever sibling still walks — helpers change walks all the time, with few helpers per walk.  With HNSW_GPU_TEAM_SPEC 0 / 1 / 2
    those few helpers take slices, so a helper's completion word meets walking waves whose job numbers coincide with the ones
    it served before (the round-3 device finding at Q = 1024 / 1536 dims: profiles/r3d_c5_spec_mismatch.txt).  Every answer ==
    the oracle's, for many launches."""
    import torch
    rng = np.random.default_rng(1234 + dim)
    port, X = build_port(n, dim, m, 64, func, k=40, seed=4000 + dim)
    ix = mirror(port, func, efs=128)
    ef = 128
    pool = gmm(4096, dim, k=40, seed=4000 + dim, stream=1)
    want_all = port.search_many(pool, ef, nthreads=8)
    dpool = torch.from_numpy(pool).cuda()
    t_end = time.time() + 25.0
    launches = 0
    for r in range(400):
        if time.time() > t_end and r >= 8:
            break
        nq = int(rng.choice([130, 300, 700, 1024, 1500]))
        o = int(rng.integers(0, 4096 - nq + 1))
        spec = str(rng.choice([0, 0, 1, 2, 2, 5]))
        _setenv(monkeypatch, {"HNSW_GPU_TEAM": "1", "HNSW_GPU_TEAM_SPEC": spec})
        out = ix.search_torch(dpool[o:o + nq], ef, stats=True)
        torch.cuda.synchronize()
        sub = {k: v[o:o + nq] for k, v in want_all.items() if isinstance(v, np.ndarray)}
        _same(out, sub, nq, (dim, r, nq, spec))
        launches += 1
    h = ix.health()
    print(f"\n[moving helpers {dim}d func {func}] {launches} launches exact; health {h}")
    assert h["slices_delivered"] > 0 and h["aborted_waves"] == 0, h
    ix.close()


@pytest.mark.timeout(300, method="thread")
def test_small_launches_on_two_streams_beside_a_big_one(monkeypatch):
    """The server's shape (csrc/server_main.cpp): 40 000 queries on stream A, meanwhile launches of 1-64 queries back to back on
    streams B and C through search contexts.  The small launches' blocks become resident one by one as the big launch's waves
    retire, so their waves take several queries each with helpers attached and detached at odd moments."""
    import torch
    _setenv(monkeypatch, {})
    dim, m, func, n, ef = 768, 16, pg.DIST_L2, 16000, 128
    port, X = build_port(n, dim, m, 64, func, k=60, seed=4242)
    ix = mirror(port, func, efs=ef)
    uniq = 2048
    Qu = gmm(uniq, dim, k=60, seed=4242, stream=1)
    want = port.search_many(Qu, ef, nthreads=8)
    reps = 20
    big = torch.from_numpy(np.tile(Qu, (reps, 1))).cuda()                      # 40 960 queries
    dQu = torch.from_numpy(Qu).cuda()
    sA, sB, sC = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    cB, cC = pg.SearchContext(ix), pg.SearchContext(ix)

    def outs(nq):
        return {"labels": torch.empty((nq, ef), dtype=torch.int64, device="cuda"), "dists": torch.empty((nq, ef), dtype=torch.float32, device="cuda"),
                "counts": torch.empty((nq,), dtype=torch.int32, device="cuda"), "stats": torch.empty((nq, 2), dtype=torch.int32, device="cuda")}

    rng = np.random.default_rng(5)
    small_checked = 0
    t_end = time.time() + 30.0
    for rnd in range(40):
        if time.time() > t_end and rnd >= 3:
            break
        torch.cuda.synchronize()
        with torch.cuda.stream(sA):
            obig = ix.search_torch(big, ef, stats=True)
        pend = []
        for k in range(24):                                    # back to back, no waiting in between
            for ctx, st in ((cB, sB), (cC, sC)):
                nq = int(rng.choice([1, 1, 2, 3, 8, 17, 64]))
                o = int(rng.integers(0, uniq - nq + 1))
                q = dQu[o:o + nq]
                out = outs(nq)
                ctx.search_torch(q, ef, out, stream=st)
                pend.append((o, nq, out))
        torch.cuda.synchronize()
        for o, nq, out in pend:
            sub = {k: v[o:o + nq] for k, v in want.items() if isinstance(v, np.ndarray)}
            _same(out, sub, nq, ("small", rnd, o, nq))
            small_checked += nq
        lab = obig["labels"].cpu().numpy().view(np.uint64).reshape(reps, uniq, ef)
        assert (lab == want["labels"][None]).all(), ("big launch", rnd)
        assert (bits(obig["dists"].cpu().numpy()).reshape(reps, uniq, ef) == bits(want["dists"])[None]).all()
    h = ix.health()
    print(f"\n[two-stream stress] {small_checked} small-launch answers + {rnd + 1} x 40 960 big-launch answers exact; health of the default workspace {h}")
    assert small_checked > 0 and h["aborted_waves"] == 0
    cB.close(); cC.close(); ix.close()


@pytest.mark.timeout(120, method="thread")
@pytest.mark.parametrize("env", [{}, {"HNSW_GPU_TEAM": "1"}, {"HNSW_GPU_BEAM": "0"}, {"HNSW_GPU_FORCE_LDS_HEAPS": "1"}])
def test_a_launch_that_is_asked_to_end_does_end(env, monkeypatch):
    """hnsw_gpu_index_abort while a long launch runs: it ends early (its waves count themselves in the health words), and the next
    launch on the same workspace — whose bitmaps the aborted waves left dirty — equals the oracle."""
    import torch
    _setenv(monkeypatch, env)
    dim, m, func, n, ef = 64, 16, pg.DIST_L2, 20000, 100
    port, X = build_port(n, dim, m, 64, func, k=40, seed=77)
    ix = mirror(port, func, efs=ef)
    Q = torch.from_numpy(gmm(1 << 16, dim, k=40, seed=77, stream=1)).cuda().repeat(32, 1)       # 2 M queries: a second or so of work
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    t0 = time.time()
    with torch.cuda.stream(side):
        big = ix.search_torch(Q, ef)
    time.sleep(0.05)
    ix.abort()
    side.synchronize()
    took = time.time() - t0
    h = ix.health()
    print(f"\n[abort {env}] kernel {ix.last_search_kernel()}: launch ended {took * 1e3:.0f} ms after it began; health {h}")
    assert h["aborted_waves"] > 0 and h["abort_pending"] == 1 and h["abort_requests"] == 1, h
    # what an interrupted launch leaves is defined per query (include/hnsw_gpu.h): a result with its count, or HNSW_GPU_COUNT_ABORTED
    cnt = big["counts"].cpu().numpy().view(np.uint32)
    unanswered = cnt == 0xFFFFFFFF
    assert unanswered.any() and (cnt[~unanswered] <= ef).all(), (int(unanswered.sum()), cnt[~unanswered].max() if (~unanswered).any() else None)
    answered = np.nonzero(~unanswered)[0]
    if len(answered):                                           # ... and the results it did deliver are results
        pick = answered[:: max(1, len(answered) // 64)][:64]
        wantb = port.search_many(Q[pick].cpu().numpy(), ef, nthreads=8)
        gotl = big["labels"].cpu().numpy().view(np.uint64)[pick]
        assert (gotl == wantb["labels"]).all(), "a query the interrupted launch answered differs from the oracle"
    Q2 = gmm(300, dim, k=40, seed=78, stream=2)
    want = port.search_many(Q2, ef, nthreads=8)
    out = ix.search_torch(torch.from_numpy(Q2).cuda(), ef, stats=True)
    torch.cuda.synchronize()
    _same(out, want, 300, "after the abort")
    assert ix.health()["abort_pending"] == 0
    ix.close()


@pytest.mark.timeout(180, method="thread")
def test_the_librarys_own_watchdog_ends_a_launch_that_runs_too_long():
    """HNSW_GPU_WATCHDOG_S (read once, at the library's first workspace — hence a process of its own): a search launch that has been
    running longer than that is asked to end by a library thread, says so on stderr, and counts the waves that left."""
    import subprocess
    import sys
    code = r'''
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from util import build_port, mirror
port, X = build_port(20000, 64, 16, 64, pg.DIST_L2, k=40, seed=77)
ix = mirror(port, pg.DIST_L2, efs=100)
Q = torch.from_numpy(gmm(1 << 16, 64, k=40, seed=77, stream=1)).cuda().repeat(32, 1)       # 2 M queries on 16 blocks (HNSW_GPU_MAX_BLOCKS): tens of seconds of work
torch.cuda.synchronize()
t0 = time.time()
ix.search_torch(Q, 100)
torch.cuda.synchronize()
print("ENDED_AFTER_S", round(time.time() - t0, 2), "HEALTH", ix.health())
'''
    env = dict(os.environ, HNSW_GPU_WATCHDOG_S="1", HNSW_GPU_MAX_BLOCKS="16")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=170, env=env,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("ENDED_AFTER_S")][0]
    secs = float(line.split()[1])
    print("\n[library watchdog] " + line)
    assert "hnsw_gpu watchdog" in r.stderr and "aborting it" in r.stderr, r.stderr[-1500:]
    assert "'aborted_waves': 0" not in line and secs < 3.5, line
