"""The kernels' OWN source on the CPU: pg_embedding_amd/csrc/hnsw_gpu.hip, gpu_*.hip and their device headers, unmodified, compiled for the
host against a SIMT emulator (tests/emu/hip/hip_runtime.h: a wavefront = an OS thread, its 64 lanes = coroutines that meet at
every cross-lane operation) and compared with the oracle bit for bit.

Test infrastructure: the product is the hipcc build for gfx950 and has no CPU path (tests/test_abi.py checks that); nothing
outside tests/ can reach the emulated library.  What it adds to the CPU tier, which otherwise only sees the oracle and the host
logic: (1) every kernel form the host can pick — beam / two-set register / LDS form, one-wave and team form, the five row
shapes, three metrics — walks and emits exactly as the oracle does; (2) the lock-free protocols between the waves of a block
run under real preemptive schedules, including a wave that walks many queries while its siblings help (on
the device: tests/test_gpu_team_stress.py), with helpers that deliver and with helpers that never do; (3) the host's abort word."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu                                           # noqa: E402

RUN = os.path.join(ROOT, "tests", "emu", "run_emu_case.py")


def run_case(case, lib, env=None, timeout=900):
    r = subprocess.run([sys.executable, RUN, case, lib], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu.build()


def test_every_kernel_form_walks_and_emits_like_the_oracle(emu_lib):
    res = run_case("forms", emu_lib)
    bad = [r for r in res if r["wrong"]]
    assert not bad, bad
    kernels = {r["kernel"] for r in res}
    for needle in ("hnsw_search_kernel_beam<0, pgemb::Shape2x2, 2, false, true>", "hnsw_search_kernel_lds<",
                   "Shape12x2, 2, true, false>", "Shape4x2", "kernel_beam<1,", "kernel_beam<2,"):
        assert any(needle in k for k in kernels), (needle, sorted(kernels))


def test_accept_decisions_that_depend_on_each_other(emu_lib):
    """Round 6's accept step decides the rows of a hop against the set as it stood at the hop's start plus the rows accepted so far in the
    hop, and appends the accepted rows in one step: beams of 1-9 over 32-link lists and quantised rows (exact ties) make those decisions
    depend on each other in every hop — ids, distance bits, counts, pop sequence length and evaluation count == oracle — and the same
    source without the in-hop term answers wrongly (the scenario has teeth)."""
    res = run_case("accept", emu_lib, timeout=1500)
    bad = [r for r in res if r["wrong"] or r["trace_wrong"]]
    assert not bad and len(res) >= 90, bad

    def edit(f, txt):
        if f == "device_search.h":
            term = " + (uint32_t) __builtin_popcountll(acc & __ballot(od_mine <= od));"
            assert txt.count(term) == 1
            txt = txt.replace(term, ";")
        return txt
    broken = build_emu.build_tree(tag="noterm", edit=edit)
    res = run_case("accept", broken, env={"EMU_ACCEPT_QUICK": "1"}, timeout=1500)
    assert sum(r["wrong"] for r in res) > 0, "the scenario does not notice a missing in-hop term"

    # the one-wave kernels take the next hop's pop from a scan made during the link-list fetch unless a row accepted in the hop beats it
    # ("Early pop"): the same source that never looks at the accepted rows pops the wrong element
    def edit2(f, txt):
        if f == "device_search.h":
            line = "nx_valid = EARLY_POP && nx_taken && !nx_beaten;"
            assert txt.count(line) == 1
            txt = txt.replace(line, "nx_valid = EARLY_POP && nx_taken;")
        return txt
    broken = build_emu.build_tree(tag="earlypop", edit=edit2)
    res = run_case("accept", broken, env={"EMU_ACCEPT_QUICK": "1"}, timeout=1500)
    assert sum(r["wrong"] + r["trace_wrong"] for r in res) > 0, "the scenario does not notice an early pop that ignores the hop's accepted rows"


def test_the_other_kernels_behind_the_c_abi(emu_lib):
    """serial device insert == the oracle's graph bytes, the walk's pop sequence, vacuum flags, a batched build that the
    search finds its way in; and the distance entry points: the device tier's own test file, unchanged, on the emulated library"""
    res = run_case("others", emu_lib)
    recall = res.pop("batched_insert_recall_at_10")
    assert all(v == 0 for v in res.values()), res
    assert recall > 0.6, recall
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_dist.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PGEMB_GPU_LIB=emu_lib))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def test_the_two_launch_insert_builds_the_oracles_graph(emu_lib):
    """device_insert.h (pair triangle by all wavefronts of several blocks, chain by the last block, one block per target): row by row
    through hnsw_gpu_index_insert_one and hnsw_gpu_index_insert_candidates the graph bytes and the returned lists are the oracle's.
    (All three functions, the general builder path beside it and a shaken schedule: tests/emu/run_emu_case.py insert with
    SIMT_EMU_JITTER=1, minutes — run when device_insert.h changes.)"""
    res = run_case("insert", emu_lib, env={"EMU_INSERT_QUICK": "1"})
    two, general = res.pop("two_launch_inserts"), res.pop("general_inserts")
    assert res and all(v == 0 for v in res.values()), res
    assert two == 300 and general == 0, (two, general)


def test_many_walks_without_the_helper_bit_clear_is_reported(capsys):
    """One wave, 32 neighbouring queries one after the other, seven helpers attached (the schedule of
    test_slice_helpers_and_hop_wide_append_are_exact_and_complete) is what the clear at the start of a walk (device_search.h) is for: without that one line a good part
    of the answers is wrong.  Reported, not asserted — it is a race, and a test must not depend on losing one."""
    CLEAR = "if (TEAM) __hip_atomic_store(&ctl[wib].helpers, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);"

    def drop_clear(name, txt):
        if name == "device_search.h":
            assert txt.count(CLEAR) == 1
            txt = txt.replace(CLEAR, ";")
        return txt
    lib = build_emu.build_tree(tag="noclear", edit=drop_clear)
    res = run_case("second_walk", lib, {"HNSW_GPU_TEAM_SPEC": "8", "EMU_SECOND_WALK_QUICK": "1"})
    with capsys.disabled():
        print("\n[simt emulator] without the helper-bit clear: " + "; ".join(f"{r['wrong']} of {r['walks']} answers wrong" for r in res))


@pytest.mark.parametrize("spec", ["5", "0"])
def test_slice_helpers_and_hop_wide_append_are_exact_and_complete(emu_lib, spec):
    """A wave that walks MANY queries with helpers attached — one wave, 32 neighbouring queries, seven helpers, with and without
    schedule jitter; two waves with three helpers each — with helpers scoring slices of its many-row hops (device_search.h, banner
    at TeamCtl) and the one-step append below ef: the many-walks schedule equals the oracle with five of seven helpers speculating (the default) and with none (all of
    them take slices; "all speculate" is what test_a_wave_that_walks_many_queries_… and the forms test run with fewer helpers);
    slices ARE delivered (the mechanism is in use) and none times out."""
    res = run_case("second_walk", emu_lib, {"HNSW_GPU_TEAM_SPEC": spec}, timeout=600)
    assert all(r["wrong"] == 0 for r in res), res
    h = res[-1]["health"]                                  # totals of the mirror's life
    assert h["slice_timeouts"] == 0 and h["aborted_waves"] == 0, h
    assert h["slices_delivered"] > 100, h


def test_a_helper_that_never_delivers_costs_time_not_answers():
    """Every inter-wave wait is bounded and falls back to the walking wave doing the work itself: the same source with the
    helpers' completion store removed (a protocol bug of the worst kind: every job is left waiting) still answers every
    query exactly as the oracle does, and the health words say what happened."""
    DONE = "if (lane == 0) __hip_atomic_store(&ctl[wib].done, js * 8u + target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);"

    def drop_done(name, txt):
        if name == "device_search.h":
            assert txt.count(DONE) == 1
            txt = txt.replace(DONE, ";")
            assert txt.count("constexpr uint32_t SLICE_WAIT_POLLS = 20000;") == 1
            txt = txt.replace("constexpr uint32_t SLICE_WAIT_POLLS = 20000;", "constexpr uint32_t SLICE_WAIT_POLLS = 50;")   # (emulated polls are slow)
        return txt
    lib = build_emu.build_tree(tag="nodone", edit=drop_done)
    res = run_case("second_walk", lib, {"HNSW_GPU_TEAM_SPEC": "0", "EMU_SECOND_WALK_QUICK": "1"}, timeout=900)
    assert all(r["wrong"] == 0 for r in res), res
    h = res[-1]["health"]
    assert h["slice_timeouts"] > 50 and h["slices_delivered"] == 0, h


def test_a_launch_that_is_asked_to_end_does_end(emu_lib):
    """hnsw_gpu_index_abort from another thread while a launch runs (every kernel form): the launch ends within the time of a few
    hops instead of minutes, the health words count the waves that left, the next launch on the same workspace is exact."""
    res = run_case("abort", emu_lib, timeout=600)
    for r in res:
        assert r["seconds_until_the_launch_ended"] < 30, r
        assert r["health_after_abort"]["aborted_waves"] > 0 and r["health_after_abort"]["abort_pending"] == 1, r
        assert r["health_after_next"]["abort_pending"] == 0 and r["wrong_after"] == 0, r
        assert "asked to end early" in r["error_of_the_interrupted_call"], r


def test_the_evaluation_trace_is_the_walk_and_the_replay_runs_over_it(emu_lib):
    """hnsw_gpu_search_traced_dev (every kernel form): the traced rows are exactly what the walk must score — the entry point, then
    the unvisited links of every popped element in order — and hnsw_gpu_replay_roof gathers exactly those bytes."""
    res = run_case("traced", emu_lib, timeout=600)
    for r in res:
        assert r["wrong"] == 0 and r["replay_rc"] == 0 and r["replay_bytes"] == r["want_bytes"] and r["word_sum_ok"], r
        assert r["parts_ok"] and r["walkers_ok"], r          # (round 4: the trace in pieces per query; walking waves per block by the caller's hint)


@pytest.mark.parametrize("env", [{"SIMT_EMU_DEVICES": "3"}, {"SIMT_EMU_DEVICES": "3", "SIMT_EMU_PEER": "0"},
                                 {"SIMT_EMU_DEVICES": "2", "HNSW_GPU_SHARDED_NO_PEER": "1"}])
def test_native_sharded_search_with_shards_on_several_emulated_devices(emu_lib, env):
    """The multi-device host code of hnsw_gpu_sharded_create / _search[_dev] (per-device streams and events, peer enabling,
    cross-device result stores or the staged peer copy, the merge on the home device) with the shards on two and three EMULATED
    devices — the first 8-GPU run must not be its first run.  The emulator gives every device its own pages and ends the process
    when a kernel touches another device's memory without peer access; launches and event records on a stream of the wrong
    device fail as HIP's do.  Peer access on (direct stores), no peer hardware, and peer access declined by the environment."""
    res = run_case("sharded", emu_lib, env, timeout=900)
    assert res["devices"] == int(env["SIMT_EMU_DEVICES"]) and len(res["cases"]) >= 1
    assert all(c["wrong"] == 0 for c in res["cases"]), res


def test_the_emulator_notices_a_store_into_another_devices_memory():
    """teeth of the test above: the same source with the peer enabling removed (but the direct-store path kept) must die on the
    first cross-device store"""
    def no_enable(name, txt):
        if name == "gpu_sharded.hip":
            needle = "const hipError_t pe = hipDeviceEnablePeerAccess(s->home, 0);"
            assert txt.count(needle) == 1
            txt = txt.replace(needle, "const hipError_t pe = hipSuccess;")
        return txt
    lib = build_emu.build_tree(tag="nopeer", edit=no_enable)
    r = subprocess.run([sys.executable, RUN, "sharded", lib], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, SIMT_EMU_DEVICES="2"))
    assert r.returncode == 86 and "without peer access" in r.stderr, (r.returncode, r.stderr[-800:])


def test_helpers_that_change_walks_inside_a_block(emu_lib):
    """several walking waves per block, slice helpers that move from a finished walk to a sibling's: exact (device counterpart
    with thousands of launches: tests/test_gpu_team_stress.py::test_helpers_that_move_from_walk_to_walk_inside_a_block)"""
    res = run_case("moving_helpers", emu_lib, timeout=900)
    assert all(r["wrong"] == 0 and ", true, " in r["kernel"] for r in res), res
    assert sum(r["health"]["slices_delivered"] for r in res) > 50, res


@pytest.mark.parametrize("jitter", [None, "3"])
def test_a_resident_launch_fed_from_the_host_answers_like_the_oracle(emu_lib, jitter):
    """Streams (include/hnsw_gpu.h) on the CPU: the emulator runs the resident launch on threads of its own — doorbell block beside the
    walking block — while this process publishes queries into the ring and polls the completion flags: ring reuse, 1 / 2 / 8 walking
    waves per block, both completion forms, stream closed and reopened, ordinary launches in between; with and without shaken wave
    schedules.  (On the device: tests/test_gpu_search.py::test_a_stream_answers_like_a_launch.)"""
    res = run_case("stream", emu_lib, env={"SIMT_EMU_JITTER": jitter, "EMU_STREAM_QUERIES": "70"} if jitter else None)
    assert len(res) == 3
    for r in res:
        assert r["wrong"] == 0 and r["wrong_in_ordinary_launches"] == 0 and r["alive_while_open"], r
        assert r["health"]["package_timeouts"] == 0 and r["health"]["slice_timeouts"] == 0, r


@pytest.mark.parametrize("stream,mailboxes", [(False, False), (True, False), (True, True)], ids=["lanes", "resident", "resident-mailboxes"])
def test_the_server_over_the_emulated_library(emu_lib, stream, mailboxes, monkeypatch):
    """hnsw_gpu_server's own source linked against the emulated library: backends' searches go through the real server — reader
    threads, dispatcher lanes with streamed completion (kernel-written flags polled while nothing else runs) or a stream session
    (lock-free producers, a RESIDENT launch of the product's kernel on emulator threads, answer threads) — into the product's C API,
    host code and kernels, all in the CPU tier: every answer equals the oracle's, a writer (delete flag) interleaved."""
    import threading
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle
    import pg_embedding_amd as pg
    import server_util as SU
    from pg_embedding_amd.datasets import gmm
    from pg_embedding_amd.server import RemoteClient, ServerProcess
    binary = SU.build_emu_server()
    if mailboxes:
        monkeypatch.setenv("PG_EMBEDDING_GPU_SHM", "1")          # the backends post their searches in shared-memory mailboxes (HGS_OP_SHM)
    dim, m, n, efs = 32, 8, 1200, 24
    X = gmm(n, dim, k=10, seed=71)
    port = oracle.PortIndex(dim, m, 40, efs, pg.DIST_L2)
    port.add(X, np.arange(n, dtype=np.uint64) + 500)
    Q = gmm(16, dim, k=10, seed=72)
    want = [port.search(q, efs)[:2] for q in Q]
    port.set_deleted(5, True)
    want_del = [port.search(q, efs)[:2] for q in Q]
    port.set_deleted(5, False)
    bad = []
    with ServerProcess(binary=binary, lanes=2, dispatchers=2, stream=stream, ring=256, shm_pollers=2 if mailboxes else None,
                       env={"SIMT_EMU_CUS": "2"}, start_timeout=60) as s:
        c0 = RemoteClient(s.socket_path)
        c0.upload(pg.make_meta(dim, m, 40, efs, pg.DIST_L2), 7, 1, port.raw(), n)

        def worker(t):
            try:
                c = RemoteClient(s.socket_path)
                for i, q in enumerate(Q):
                    lab, dst = c.search(7, q, efs)
                    ok = any((lab == w[0]).all() and (dst.view(np.uint32) == w[1].view(np.uint32)).all() for w in (want[i], want_del[i]))
                    if not ok:
                        bad.append((t, i))
                c.close()
            except Exception as ex:            # noqa: BLE001
                bad.append((t, repr(ex)))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
        [t.start() for t in th]
        for i in range(6):
            c0.set_deleted(7, 5, i % 2 == 0)
        [t.join() for t in th]
        st = c0.stats()
        c0.close()
    assert not bad, bad
    assert st["searches"] == 4 * len(Q) and st["search_errors"] == 0, st
    if stream:
        # sessions, not batches — except while the interleaved writer's control work holds the device: since round 6 the searches that are
        # waiting then are served by ordinary blocking launches instead of waiting for all of it (server_main.cpp, "control work first"),
        # so a batch of at most the four backends may have formed
        assert st["max_batch"] <= len(th) and st["batches"] >= 1, st
    assert st["shm_searches"] == (st["searches"] if mailboxes else 0), st


def test_wide_beam_form_equals_the_oracle(emu_lib):
    """device_search_wide.h on the CPU: every beam from 1 to beyond the index size, ties, vacuumed rows, the pop sequence"""
    res = run_case("wide", emu_lib, timeout=900)
    assert all(r["wrong"] == 0 and r["trace_wrong"] == 0 and "kernel_wide" in r["kernel"] for r in res), res


def test_reference_order_arithmetic_returns_the_compiled_references_ids(emu_lib):
    """HNSW_GPU_REF_ORDER=1 (debug): distances summed in the order of oracle/_ref's own build of distfunc.c — the kernels' id lists
    and distance bits then equal the COMPILED REFERENCE's for every query, with no canonical-order oracle in between."""
    res = run_case("reforder", emu_lib, timeout=900)
    if isinstance(res, dict):
        pytest.skip(res["skipped"])
    ran = [r for r in res if "skipped" not in r]
    if not ran:
        pytest.skip("this host's _ref build sums in another order than the one score_rows_ref restates")
    assert all(any(f"kernel_beam<{c}" in r["kernel"] for c in (3, 4, 5)) for r in ran), res      # L2 / Manhattan / cosine in the reference build's order
    assert all(r["wrong_vs_the_compiled_reference"] == 0 for r in ran), res
    assert {r["func"] for r in ran} == {0, 1, 2}, res
