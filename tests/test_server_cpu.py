"""hnsw_gpu_server + libembedding_gpuc.so without a GPU: the server's own source linked against the
oracle-backed engine double (tests/double/engine_double.c).  What is under test here is the part
that has no arithmetic in it — wire protocol, batching, per-mirror locking, generations, the
client library's drop-in symbols and their failure behaviour; the device path behind the same
server is covered by tests/test_gpu_server.py."""
import json
import os
import socket
import struct
import subprocess
import threading
import time

import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from pg_embedding_amd.server import HGS_ERR_NOKEY, HGS_ERR_STALE, RemoteClient, RemoteError, ServerProcess
import server_util as SU

os.environ["PG_EMBEDDING_GPU_SHM"] = "1"      # the backends of this module post their searches in mailboxes when the server polls them


@pytest.fixture(scope="module")
def double_bin():
    return SU.build_double_server()


@pytest.fixture(params=[(2, False, 2), (0, False, 2), (2, True, 2), (2, False, 0)], ids=["streamed", "blocking", "resident", "socket-only"])
def srv(double_bin, request):
    """The dispatcher forms: streamed completion (2 launches in flight per dispatcher, answers leave as
    their walks end), one blocking launch at a time, and --stream 1: searches through one resident launch
    per (mirror, efsearch) fed through a ring (the engine double plays the launch with threads that take
    tickets and answer ring slots out of order, tests/double/engine_double.c) — the smallest ring the server accepts (256 slots)."""
    lanes, stream, pollers = request.param
    # (searches travel through the backends' shared-memory mailboxes, HGS_OP_SHM: --shm-pollers 2 + PG_EMBEDDING_GPU_SHM=1 above;
    # "socket-only" = the server's default, no pollers: it refuses mailboxes and every request stays on the socket)
    with ServerProcess(binary=double_bin, lanes=lanes, stream=stream, ring=256, shm_pollers=pollers) as s:
        yield s


def port_index(n, dim, m, efc, efs, func, seed):
    X = gmm(n, dim, k=20, seed=seed)
    p = oracle.PortIndex(dim, m, efc, efs, func)
    p.add(X, np.arange(n, dtype=np.uint64) + 500)
    return p, X


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_product_server_refuses_to_start_without_a_device():
    """No device, no service: the shipped binary has no CPU engine behind it."""
    from pg_embedding_amd._lib import gpu_lib
    if gpu_lib().hnsw_gpu_device_count() > 0:
        pytest.skip("a device is present")
    with pytest.raises(RuntimeError, match="status 3"):
        ServerProcess().start()


def test_client_fails_loudly_without_a_server(tmp_path):
    with pytest.raises(RemoteError) as e:
        RemoteClient(str(tmp_path / "nobody-listens"))
    assert e.value.code == -23
    exe = SU.build_c_client("dropin_demo")
    env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=str(tmp_path / "nobody-listens"))
    r = subprocess.run([exe, "50", "8", "3", "8", "8", "2"], capture_output=True, text=True, env=env)
    assert r.returncode != 0 and "cannot reach hnsw_gpu_server" in r.stderr


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
def test_remote_calls_round_trip(srv, func):
    dim, m, n, efs = 40, 6, 1200, 48
    port, X = port_index(n, dim, m, 32, efs, func, seed=11 + func)
    meta = pg.make_meta(dim, m, 32, efs, func)
    c = RemoteClient(srv.socket_path)
    key = 4242
    assert c.lookup(key) == (False, 0, 0)
    with pytest.raises(RemoteError) as e:
        c.search(key, X[0], efs)
    assert e.value.code == HGS_ERR_NOKEY
    c.upload(meta, key, 7, port.raw(), n)
    assert c.lookup(key) == (True, 7, n)
    Q = gmm(25, dim, k=20, seed=11 + func, stream=1)
    for q in Q:
        lab, dst = c.search(key, q, efs)
        wl, wd = port.search(q, efs)[:2]
        assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
    # a generation the server does not hold is refused, 0 means "whatever is current"
    with pytest.raises(RemoteError) as e:
        c.search(key, Q[0], efs, gen=8)
    assert e.value.code == HGS_ERR_STALE
    assert (c.search(key, Q[0], efs, gen=7)[0] == port.search(Q[0], efs)[0]).all()
    # other beams, including one wider than... the doubling of embedding.c:334
    for ef in (1, 5, 2 * efs):
        assert (c.search(key, Q[1], ef)[0] == port.search(Q[1], ef)[0]).all()
    # vacuum flag (embedding.c:920-926)
    first = int(c.search(key, Q[2], efs)[0][0])
    idx = first - 500
    c.set_deleted(key, idx, True)
    port.set_deleted(idx, True)
    assert first not in c.search(key, Q[2], efs)[0].tolist()
    assert (c.search(key, Q[2], efs)[0] == port.search(Q[2], efs)[0]).all()
    # incremental update: new rows linked by the host, only the changed elements travel
    more = gmm(30, dim, k=20, seed=99)
    before = port.raw().reshape(n, -1).copy()
    port.add(more, np.arange(30, dtype=np.uint64) + 9000)
    after = port.raw().reshape(n + 30, -1)
    changed = np.flatnonzero((after[:n] != before).any(axis=1))
    with pytest.raises(RemoteError) as e:
        c.update(meta, key, 6, 8, after[n:].reshape(-1), n, 30)       # wrong expected generation
    assert e.value.code == HGS_ERR_STALE
    c.update(meta, key, 7, 8, after[n:].reshape(-1), n, 30)
    for i in changed:
        c.update(meta, key, 8, 8, after[i], int(i), 1)
    assert c.lookup(key) == (True, 8, n + 30)
    for q in Q[:10]:
        assert (c.search(key, q, efs)[0] == port.search(q, efs)[0]).all()
    # the mirror comes back as the host's element images
    img = c.export(key, (n + 30) * meta.size_data_per_element)
    assert (img == port.raw()).all()
    st = c.stats()
    assert st["mirrors"] == 1 and st["mirror_elements"] == n + 30 and st["uploads"] == 1 and st["search_errors"] == 0
    c.drop(key)
    assert c.lookup(key)[0] is False
    with pytest.raises(RemoteError):
        c.drop(key)
    c.close()


def test_bulk_link_on_the_server_matches_serial_inserts(srv):
    """UPLOAD of zero-linked rows + LINK (the CREATE INDEX offload) in serial mode = the oracle's graph."""
    dim, m, n = 16, 4, 300
    port, X = port_index(n, dim, m, 16, 20, pg.DIST_L2, seed=3)
    meta = pg.make_meta(dim, m, 16, 20, pg.DIST_L2)
    img = port.raw().reshape(n, -1).copy()
    img[:, :meta.offset_data] = 0                         # forget the links
    c = RemoteClient(srv.socket_path)
    c.upload(meta, 9, 1, img.reshape(-1), n)
    c.link(9, 0, n, 1)
    assert (c.export(9, n * meta.size_data_per_element) == port.raw()).all()
    c.close()


def run_clients(sock, key, gen, dim, m, efc, efs, func, Q, nproc, tmp_path, rounds=1):
    exe = SU.build_c_client("server_clients")
    qf, of = str(tmp_path / "q.f32"), str(tmp_path / "out.u64")
    np.ascontiguousarray(Q, np.float32).tofile(qf)
    env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=sock)
    r = subprocess.run([exe, str(key), str(gen), str(dim), str(m), str(efc), str(efs), str(func), qf, str(len(Q)),
                        str(nproc), of, str(rounds)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr
    out = np.fromfile(of, np.uint64)
    nq = len(Q)
    return json.loads(r.stdout), out[:nq * efs].reshape(nq, efs), out[nq * efs:]


@pytest.mark.parametrize("lanes", [2, 0])
def test_many_backends_are_batched_and_each_gets_its_own_answer(double_bin, tmp_path, lanes):
    """48 single-threaded processes call hnsw_search() one query at a time (embedding.c:317); the
    server turns what is waiting into batches; every call returns exactly the oracle's array."""
    dim, m, n, efs = 32, 5, 2000, 24
    port, X = port_index(n, dim, m, 24, efs, pg.DIST_L2, seed=21)
    meta = pg.make_meta(dim, m, 24, efs, pg.DIST_L2)
    Q = gmm(960, dim, k=20, seed=21, stream=1)
    with ServerProcess(binary=double_bin, lanes=lanes, env={"HGS_DOUBLE_SLEEP_US": "3000"}) as s:
        c = RemoteClient(s.socket_path)
        c.upload(meta, 77, 3, port.raw(), n)
        info, labels, counts = run_clients(s.socket_path, 77, 3, dim, m, 24, efs, pg.DIST_L2, Q, 48, tmp_path)
        want = port.search_many(Q, efs)
        assert (counts == want["counts"]).all()
        for q in range(len(Q)):
            k = int(counts[q])
            assert (labels[q, :k] == want["labels"][q, :k]).all()
        st = c.stats()
        assert st["searches"] == len(Q) and st["search_errors"] == 0
        assert st["max_batch"] > 8 and st["batches"] < len(Q) // 4, st       # really coalesced
        assert st["connections"] >= 49
        c.close()


@pytest.mark.parametrize("walkers", [None, 1])
def test_many_backends_through_a_resident_launch(double_bin, tmp_path, walkers):
    """--stream 1: 48 single-threaded processes call hnsw_search() one query at a time; reader threads write ring slots without a
    lock, the launch's walking threads answer them out of order, answer threads hand each backend its own result: every call returns
    exactly the oracle's array, through a ring (256 slots) far smaller than the number of queries; sessions are opened, re-shaped by
    load (walkers auto) and closed when idle — several per run."""
    dim, m, n, efs = 32, 5, 2000, 24
    port, X = port_index(n, dim, m, 24, efs, pg.DIST_L2, seed=23)
    meta = pg.make_meta(dim, m, 24, efs, pg.DIST_L2)
    Q = gmm(960, dim, k=20, seed=23, stream=1)
    with ServerProcess(binary=double_bin, lanes=2, stream=True, ring=256, walkers=walkers, shm_pollers=2 if walkers else None,
                       env={"HGS_DOUBLE_SLEEP_US": "2000"}) as s:
        c = RemoteClient(s.socket_path)
        c.upload(meta, 78, 3, port.raw(), n)
        info, labels, counts = run_clients(s.socket_path, 78, 3, dim, m, 24, efs, pg.DIST_L2, Q, 48, tmp_path)
        want = port.search_many(Q, efs)
        assert (counts == want["counts"]).all()
        for q in range(len(Q)):
            k = int(counts[q])
            assert (labels[q, :k] == want["labels"][q, :k]).all()
        time.sleep(0.05)                                            # the idle session closes (2 ms) and is counted
        st = c.stats()
        assert st["searches"] == len(Q) and st["search_errors"] == 0, st
        assert st["max_batch"] == 0 and 1 <= st["batches"] < len(Q) // 8, st      # no batch was formed: sessions, a few of them
        c.close()


def test_a_server_whose_device_has_no_streams_falls_back_to_launches(double_bin, tmp_path):
    """--stream 1 on a device that cannot open a stream (the double with HGS_DOUBLE_NO_STREAMS; on hardware: efsearch > 512): the
    manager answers through ordinary launches, same arrays."""
    dim, m, n, efs = 24, 4, 900, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=27)
    Q = gmm(96, dim, k=20, seed=27, stream=1)
    with ServerProcess(binary=double_bin, lanes=2, stream=True, ring=256, env={"HGS_DOUBLE_NO_STREAMS": "1"}) as s:
        c = RemoteClient(s.socket_path)
        c.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 79, 1, port.raw(), n)
        for q in Q:
            lab, dst = c.search(79, q, efs)
            wl, wd = port.search(q, efs)[:2]
            assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
        st = c.stats()
        assert st["searches"] == len(Q) and st["search_errors"] == 0 and st["max_batch"] >= 1, st
        c.close()


@pytest.mark.parametrize("stream", [False, True], ids=["lanes", "resident"])
def test_the_servers_threads_are_race_free(tmp_path, stream):
    """The server's own source under ThreadSanitizer (every object of the double link built with -fsanitize=thread): twelve client
    threads with two beams search while a writer toggles a delete flag — readers, dispatcher lanes or lock-free stream producers /
    answer threads / session manager, the mirror's gate, the control thread and the double's device threads: no report.  (The
    searches' answers are checked too: each equals the oracle's for one of the two states of the flag.)"""
    import glob
    binary = SU.build_double_server_tsan()
    dim, m, n, efs = 24, 4, 900, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=51)
    Q = gmm(30, dim, k=20, seed=51, stream=1)
    victim = 3
    want = {}
    for ef in (8, efs):
        a = [port.search(q, ef)[0] for q in Q]
        port.set_deleted(victim, True)
        b = [port.search(q, ef)[0] for q in Q]
        port.set_deleted(victim, False)
        want[ef] = (a, b)
    log = str(tmp_path / "tsan")
    env = {"TSAN_OPTIONS": f"log_path={log} exitcode=0 halt_on_error=0", "HGS_DOUBLE_SLEEP_US": "500"}
    bad = []
    with ServerProcess(binary=binary, lanes=2, stream=stream, ring=256, shm_pollers=2, env=env) as s:
        c0 = RemoteClient(s.socket_path)
        c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 5, 1, port.raw(), n)

        def worker(t):
            try:
                c = RemoteClient(s.socket_path)
                ef = efs if t % 2 else 8
                for i, q in enumerate(Q):
                    lab = c.search(5, q, ef)[0]
                    if not (np.array_equal(lab, want[ef][0][i]) or np.array_equal(lab, want[ef][1][i])):
                        bad.append((t, i))
                c.close()
            except Exception as ex:            # noqa: BLE001
                bad.append((t, repr(ex)))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
        [t.start() for t in th]
        for i in range(40):
            c0.set_deleted(5, victim, i % 2 == 0)
        [t.join() for t in th]
        st = c0.stats()
        c0.close()
    assert not bad, bad
    assert st["searches"] == 12 * len(Q) and st["search_errors"] == 0, st
    reports = "".join(open(f).read() for f in glob.glob(log + "*"))
    assert "ThreadSanitizer" not in reports, reports[:4000]


def test_searches_travel_through_mailboxes_and_fall_back_to_the_socket(double_bin, tmp_path):
    """HGS_OP_SHM (include/hnsw_gpu_server.h): a connection's searches are posted in its shared-memory mailbox — the server counts them
    (hgs_stats.shm_searches) — and equal the socket's answers bit for bit; what does not fit a mailbox (a beam of more than 1 024
    results here), a client that did not ask (PG_EMBEDDING_GPU_SHM unset or 0, separate processes) and a server without pollers (the
    default) go through the socket, same arrays."""
    dim, m, n, efs = 24, 4, 900, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=61)
    meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)
    Q = gmm(48, dim, k=20, seed=61, stream=1)
    for pollers in (2, None):
        with ServerProcess(binary=double_bin, lanes=2, shm_pollers=pollers) as s:
            c = RemoteClient(s.socket_path)
            c.upload(meta, 9, 1, port.raw(), n)
            for q in Q:
                for ef in (efs, 5):
                    lab, dst = c.search(9, q, ef)
                    wl, wd = port.search(q, ef)[:2]
                    assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
            st = c.stats()
            assert st["searches"] == 2 * len(Q) and st["shm_searches"] == (2 * len(Q) if pollers else 0), st
            lab, dst = c.search(9, Q[0], 1500)                      # more results than the mailbox holds: this one on the socket
            wl, wd = port.search(Q[0], 1500)[:2]
            assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
            assert c.stats()["shm_searches"] == st["shm_searches"]
            if pollers:
                os.environ["PG_EMBEDDING_GPU_SHM"] = "0"            # (read when a connection is made: the C clients below make their own)
                try:
                    info, labels, counts = run_clients(s.socket_path, 9, 1, dim, m, 16, efs, pg.DIST_L2, Q, 4, tmp_path)
                finally:
                    os.environ["PG_EMBEDDING_GPU_SHM"] = "1"
                want = port.search_many(Q, efs)
                assert (counts == want["counts"]).all() and all((labels[i, :int(counts[i])] == want["labels"][i, :int(counts[i])]).all() for i in range(len(Q)))
                assert c.stats()["shm_searches"] == st["shm_searches"]
                info, labels, counts = run_clients(s.socket_path, 9, 1, dim, m, 16, efs, pg.DIST_L2, Q, 4, tmp_path)
                assert (counts == want["counts"]).all() and all((labels[i, :int(counts[i])] == want["labels"][i, :int(counts[i])]).all() for i in range(len(Q)))
                assert c.stats()["shm_searches"] == st["shm_searches"] + len(Q)
            c.close()


def test_a_backend_waiting_at_its_mailbox_notices_that_the_server_died(double_bin):
    """A search posted in the mailbox sleeps in futex_wait; the server is killed meanwhile: the backend finds the socket closed at
    its next look (100 ms), and fails (no server to reconnect to) instead of waiting for its time-out."""
    dim, m, n, efs = 24, 4, 900, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=63)
    s = ServerProcess(binary=double_bin, lanes=0, shm_pollers=2, env={"HGS_DOUBLE_SLEEP_US": "999999"}).start()
    try:
        c = RemoteClient(s.socket_path)
        c.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 3, 1, port.raw(), n)
        c.search(3, X[0], efs)                                       # the mailbox is set up and works
        out = {}

        def call():
            t0 = time.time()
            try:
                for i in range(200):                                 # keeps searching until the server is gone
                    c.search(3, X[i], efs)
                out["result"] = "answered"
            except RemoteError as ex:
                out["result"] = repr(ex)
            out["seconds"] = time.time() - t0

        th = threading.Thread(target=call)
        th.start()
        time.sleep(0.3)
        s.proc.kill()
        th.join(20)
        assert not th.is_alive() and out.get("result", "").startswith("RemoteError"), out
    finally:
        s.stop()


@pytest.mark.parametrize("stream,pollers", [(False, None), (True, 2)], ids=["lanes", "resident-mailboxes"])
def test_the_servers_memory_use_is_clean(stream, pollers):
    """The server's own source under AddressSanitizer + UndefinedBehaviorSanitizer: connections come and go while others search
    (mailboxes registered and unregistered under the pollers' feet), two beams, a writer that closes stream sessions: no report
    (a report ends the server with status 77), every request answered."""
    binary = SU.build_double_server_asan()
    dim, m, n, efs = 24, 4, 900, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=71)
    Q = gmm(20, dim, k=20, seed=71, stream=1)
    env = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=77", "HGS_DOUBLE_SLEEP_US": "300"}
    s = ServerProcess(binary=binary, lanes=2, stream=stream, ring=256, shm_pollers=pollers, env=env).start()
    errs = []
    try:
        c0 = RemoteClient(s.socket_path)
        c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 5, 1, port.raw(), n)

        def worker(t):
            try:
                for _ in range(3):
                    c = RemoteClient(s.socket_path)
                    for q in Q:
                        c.search(5, q, efs if t % 2 else 8)
                    c.close()
            except Exception as ex:            # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=worker, args=(t,)) for t in range(8)]
        [t.start() for t in th]
        for i in range(30):
            c0.set_deleted(5, 3, i % 2 == 0)
        [t.join() for t in th]
        st = c0.stats()
        c0.close()
    finally:
        rc = s.stop()
    assert not errs, errs
    assert rc == 0, f"the server ended with status {rc} (77 = a sanitizer report, see its stderr)"
    assert st["searches"] == 8 * 3 * len(Q) and st["search_errors"] == 0, st
    assert st["shm_searches"] == (st["searches"] if pollers else 0), st


def test_batches_never_mix_beams_or_mirrors(srv):
    """Concurrent searches with different efSearch on two mirrors: batches are per (mirror, ef)."""
    dim, m, efs = 24, 4, 16
    pa, Xa = port_index(900, dim, m, 16, efs, pg.DIST_L2, seed=31)
    pb, Xb = port_index(700, dim, m, 16, efs, pg.DIST_COSINE, seed=32)
    c0 = RemoteClient(srv.socket_path)
    c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 1, 1, pa.raw(), 900)
    c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_COSINE), 2, 1, pb.raw(), 700)
    Q = gmm(64, dim, k=20, seed=33, stream=1)
    errors = []

    def worker(t):
        try:
            c = RemoteClient(srv.socket_path)          # connections are per thread
            key, port = (1, pa) if t % 2 == 0 else (2, pb)
            ef = (8, 16, 40)[t % 3]
            for q in Q:
                lab, dst = c.search(key, q, ef)
                wl, wd = port.search(q, ef)[:2]
                if not ((lab == wl).all() and (bits(dst) == bits(wd)).all()):
                    errors.append((t, "mismatch"))
            c.close()
        except Exception as ex:            # noqa: BLE001
            errors.append((t, repr(ex)))

    th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert c0.stats()["searches"] == 12 * len(Q)
    c0.close()


def test_searches_and_mirror_changes_interleave_safely(srv):
    """Writers (SET_DELETED toggles under the mirror's write lock) while readers search: every answer
    equals the oracle's answer for one of the two states."""
    dim, m, n, efs = 24, 4, 800, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=41)
    meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)
    c0 = RemoteClient(srv.socket_path)
    c0.upload(meta, 5, 1, port.raw(), n)
    q = X[10] + 0.01
    alive = port.search(q, efs)[0]
    victim = int(alive[0]) - 500
    port.set_deleted(victim, True)
    dead = port.search(q, efs)[0]
    stop = threading.Event()
    bad = []

    def reader():
        c = RemoteClient(srv.socket_path)
        while not stop.is_set():
            got = c.search(5, q, efs)[0]
            if not (np.array_equal(got, alive) or np.array_equal(got, dead)):
                bad.append(got)
        c.close()

    th = [threading.Thread(target=reader) for _ in range(6)]
    [t.start() for t in th]
    for i in range(200):
        c0.set_deleted(5, victim, i % 2 == 0)
    stop.set()
    [t.join() for t in th]
    assert not bad
    c0.close()


def test_dropin_symbols_from_c_match_the_reference(srv):
    """The C host of tests/dropin_c (storage callbacks + the four symbols only) linked against
    libembedding_gpuc.so: inserts through hnsw_bind_point (BIND + write-back), searches through
    hnsw_search, same bytes on stdout as the same host linked against the reference's objects —
    un-attached (throw-away mirrors per call) and attached to one server-side mirror."""
    exe = SU.build_c_client("dropin_demo")
    ref = SU.build_c_reference("dropin_demo")
    args = ["300", "24", "4", "16", "12", "15"]
    env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=srv.socket_path)
    plain = subprocess.run([exe] + args, capture_output=True, text=True, env=env, check=True).stdout
    attached = subprocess.run([exe] + args + ["123"], capture_output=True, text=True, env=env, check=True).stdout
    assert plain.count("\n") == 15 and "d0=0.000000" in plain
    assert attached == plain
    if ref:
        want = subprocess.run([ref] + args, capture_output=True, text=True, check=True).stdout
        assert plain == want
    c = RemoteClient(srv.socket_path)
    st = c.stats()
    assert st["binds"] >= 2 * 299                       # both runs inserted through the server
    assert c.lookup(123) == (True, 1, 300)              # the attached run left its mirror behind, complete
    assert st["mirrors"] == 1                           # the throw-away mirrors are gone
    c.close()


@pytest.mark.parametrize("lanes", [2, 0])
def test_a_failing_batch_is_reported_and_the_server_goes_on(double_bin, lanes):
    dim, m, n = 16, 4, 200
    port, X = port_index(n, dim, m, 16, 10, pg.DIST_L2, seed=51)
    with ServerProcess(binary=double_bin, lanes=lanes, env={"HGS_DOUBLE_FAIL_EF": "13"}) as s:
        c = RemoteClient(s.socket_path)
        c.upload(pg.make_meta(dim, m, 16, 10, pg.DIST_L2), 1, 1, port.raw(), n)
        with pytest.raises(RemoteError) as e:
            c.search(1, X[0], 13)
        assert e.value.code == -4                       # HNSW_GPU_ERR_INTERNAL passed through
        assert (c.search(1, X[0], 10)[0] == port.search(X[0], 10)[0]).all()
        assert c.stats()["search_errors"] == 1
        c.close()


def test_malformed_traffic_only_costs_the_sender_its_connection(srv):
    dim, m, n = 16, 4, 200
    port, X = port_index(n, dim, m, 16, 10, pg.DIST_L2, seed=61)
    c = RemoteClient(srv.socket_path)
    c.upload(pg.make_meta(dim, m, 16, 10, pg.DIST_L2), 1, 1, port.raw(), n)

    def raw():
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.settimeout(5)
        s.connect(srv.socket_path)
        return s

    hdr = struct.Struct("<IHhIIQQQQ")
    assert hdr.size == 48
    s = raw()                                            # wrong magic
    s.sendall(b"\x00" * 48)
    assert s.recv(64) == b""
    s = raw()                                            # absurd length
    s.sendall(hdr.pack(0x31534748, 5, 0, 0xFFFFFFF0, 10, 1, 0, 0, 0))
    assert s.recv(64) == b""
    s = raw()                                            # unknown op: an error response, then closed
    s.sendall(hdr.pack(0x31534748, 99, 0, 0, 0, 0, 0, 0, 0))
    r = hdr.unpack(s.recv(48))
    assert r[2] == -20
    s = raw()                                            # wrong query size: refused, connection stays usable
    s.sendall(hdr.pack(0x31534748, 5, 0, 8, 10, 1, 0, 0, 0) + b"\x00" * 8)
    assert hdr.unpack(s.recv(48))[2] == -2
    s.sendall(hdr.pack(0x31534748, 2, 0, 0, 0, 1, 0, 0, 0))
    r = hdr.unpack(s.recv(48))
    assert r[2] == 0 and r[7] == n
    s = raw()                                            # UPLOAD that promises a descriptor and sends none
    s.sendall(hdr.pack(0x31534748, 3, 0, 88, 0, 9, 1, 5, 0) + b"\x00" * 88)
    assert hdr.unpack(s.recv(48))[2] == -20
    # an UPLOAD whose file could still shrink under the server's mapping (no F_SEAL_SHRINK) is refused
    fd = os.memfd_create("unsealed", os.MFD_CLOEXEC)
    os.ftruncate(fd, 5 * (9 * 4 + dim * 4 + 8))
    meta = pg.make_meta(dim, m, 16, 10, pg.DIST_L2)
    s = raw()
    socket.send_fds(s, [hdr.pack(0x31534748, 3, 0, 88, 0, 9, 1, 5, 0) + bytes(meta)], [fd])
    assert hdr.unpack(s.recv(48))[2] == -20
    os.close(fd)
    assert c.lookup(9)[0] is False
    # a connection may pipeline a little (a backend has one scan in flight), not flood the queue
    s = raw()
    one = hdr.pack(0x31534748, 5, 0, dim * 4, 10, 1, 0, 0, 0) + X[0].tobytes()
    s.sendall(one * 8)
    got = b""
    while len(got) < 8 * (48 + 10 * 8):
        got += s.recv(65536)
    assert all(hdr.unpack(got[i * 128:i * 128 + 48])[2] == 0 for i in range(8))
    s = raw()
    s.settimeout(10)
    try:
        s.sendall(one * 400)
        data = b"x"
        while data:
            data = s.recv(65536)                         # ... answers, then EOF: cut off
    except (BrokenPipeError, ConnectionResetError):
        pass
    s = raw()                                            # a backend that dies mid-request
    s.sendall(hdr.pack(0x31534748, 5, 0, dim * 4, 10, 1, 0, 0, 0) + b"\x00" * 10)
    s.close()
    assert (c.search(1, X[3], 10)[0] == port.search(X[3], 10)[0]).all()     # everybody else is fine
    c.close()


def test_several_servers_one_per_gpu_backends_spread_by_pid(double_bin, tmp_path):
    """PG_EMBEDDING_GPU_SERVER = "sockA,sockB": one server per GPU, every backend sticks to one of them (pid
    modulo), each server mirrors what its backends use — replicas.  All answers are still the oracle's."""
    dim, m, n, efs = 24, 4, 1200, 16
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=88)
    meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)
    Q = gmm(640, dim, k=20, seed=88, stream=1)
    with ServerProcess(binary=double_bin) as a, ServerProcess(binary=double_bin) as b:
        for s in (a, b):                                   # what the first attach of a backend on each would do
            c = RemoteClient(s.socket_path)
            c.upload(meta, 5, 1, port.raw(), n)
            c.close()
        both = a.socket_path + "," + b.socket_path
        info, labels, counts = run_clients(both, 5, 1, dim, m, 16, efs, pg.DIST_L2, Q, 32, tmp_path)
        want = port.search_many(Q, efs)
        assert (counts == want["counts"]).all()
        for q in range(len(Q)):
            assert (labels[q, :int(counts[q])] == want["labels"][q, :int(counts[q])]).all()
        served = []
        for s in (a, b):
            c = RemoteClient(s.socket_path)
            served.append(c.stats()["searches"])
            c.close()
        assert sum(served) == len(Q) and min(served) > 0, served      # both took part
        # a DROP (what VACUUM sends) must reach every replica, not only the server this process talks to
        c = RemoteClient(both)
        c.drop(5)
        c.close()
        for s in (a, b):
            c = RemoteClient(s.socket_path)
            assert c.lookup(5)[0] is False
            c.close()


def test_an_upload_that_does_not_fit_evicts_idle_mirrors_lru_first(double_bin):
    dim, m, efs = 16, 4, 10
    meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)
    ports = [port_index(400, dim, m, 16, efs, pg.DIST_L2, seed=70 + i) for i in range(4)]
    with ServerProcess(binary=double_bin, env={"HGS_DOUBLE_CAPACITY": "1000"}) as s:
        c = RemoteClient(s.socket_path)
        c.upload(meta, 1, 1, ports[0][0].raw(), 400)
        c.upload(meta, 2, 1, ports[1][0].raw(), 400)
        c.search(1, ports[0][1][0], efs)                 # key 1 is now the more recently used one
        c.upload(meta, 3, 1, ports[2][0].raw(), 400)     # 1200 > 1000: key 2 has to go
        assert [c.lookup(k)[0] for k in (1, 2, 3)] == [True, False, True]
        assert c.stats()["evictions"] == 1
        # a new generation of a key that is already there replaces it without touching the others
        c.upload(meta, 3, 2, ports[3][0].raw(), 400)
        assert [c.lookup(k)[:2] for k in (1, 3)] == [(True, 1), (True, 2)] and c.stats()["evictions"] == 1
        assert (c.search(3, ports[3][1][5], efs)[0] == ports[3][0].search(ports[3][1][5], efs)[0]).all()
        # something that cannot fit at all: refused (HNSW_GPU_ERR_NOMEM), everything idle was given up for it
        big = port_index(1100, dim, m, 16, efs, pg.DIST_L2, seed=99)[0]
        with pytest.raises(RemoteError) as e:
            c.upload(meta, 9, 1, big.raw(), 1100)
        assert e.value.code == -3
        c.close()


@pytest.mark.parametrize("lanes", [3, 0])
def test_linger_gathers_a_batch_before_launching(double_bin, lanes):
    """--linger-us N --min-batch M: a dispatcher that finds fewer than M requests waits up to N us for more."""
    dim, m, n, efs = 16, 4, 400, 10
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=97)
    with ServerProcess(binary=double_bin, lanes=lanes, linger_us=200000, min_batch=6) as s:
        c0 = RemoteClient(s.socket_path)
        c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 1, 1, port.raw(), n)
        bad = []

        def one(i):
            c = RemoteClient(s.socket_path)
            if not (c.search(1, X[i], efs)[0] == port.search(X[i], efs)[0]).all():
                bad.append(i)
            c.close()

        th = [threading.Thread(target=one, args=(i,)) for i in range(6)]
        [t.start() for t in th]
        [t.join() for t in th]
        st = c0.stats()
        assert not bad and st["searches"] == 6
        assert st["max_batch"] >= 4 and st["batches"] <= 3, st       # gathered, not six launches of one
        c0.close()


def test_a_restarted_server_is_picked_up_again(double_bin, tmp_path):
    """The server goes away and comes back on the same socket (its mirrors are gone with it): the next
    request of a connected client reconnects by itself; a search on a mirror the new server does not
    have says so (NOKEY) and works again after the upload."""
    dim, m, n, efs = 16, 4, 300, 10
    port, X = port_index(n, dim, m, 16, efs, pg.DIST_L2, seed=95)
    meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)
    path = str(tmp_path / "sock")
    s1 = ServerProcess(socket_path=path, binary=double_bin).start()
    c = RemoteClient(path)
    c.upload(meta, 1, 1, port.raw(), n)
    assert (c.search(1, X[0], efs)[0] == port.search(X[0], efs)[0]).all()
    s1.stop()
    with pytest.raises(RemoteError) as e:                 # nobody there at all: an I/O error, not a hang
        c.stats()
    assert e.value.code == -23
    s2 = ServerProcess(socket_path=path, binary=double_bin).start()
    try:
        assert c.stats()["mirrors"] == 0                  # reconnected without being told to
        with pytest.raises(RemoteError) as e:
            c.search(1, X[0], efs)
        assert e.value.code == HGS_ERR_NOKEY
        c.upload(meta, 1, 1, port.raw(), n)
        assert (c.search(1, X[0], efs)[0] == port.search(X[0], efs)[0]).all()
        # and with the connection still open on the client side when the server is replaced
        s2.stop()
        s2 = ServerProcess(socket_path=path, binary=double_bin).start()
        assert c.lookup(1) == (False, 0, 0)               # first call after the restart: one silent retry
    finally:
        s2.stop()
        c.close()


def test_server_stops_cleanly_on_sigterm(double_bin):
    s = ServerProcess(binary=double_bin).start()
    c = RemoteClient(s.socket_path)
    assert c.stats()["connections_now"] == 1
    assert s.stop() == 0
    with pytest.raises(RemoteError):
        c.stats()
    c.close()


def _sealed_memfd(data: bytes) -> int:
    import fcntl
    fd = os.memfd_create("images", os.MFD_CLOEXEC | os.MFD_ALLOW_SEALING)
    os.write(fd, data)
    fcntl.fcntl(fd, fcntl.F_ADD_SEALS, fcntl.F_SEAL_SHRINK | fcntl.F_SEAL_GROW | fcntl.F_SEAL_WRITE)
    return fd


def test_a_snapshot_cannot_replace_a_mirror_that_changed_since_its_lookup(srv):
    """a scan's walk can overlap an insert in another backend; its UPLOAD must not replace the
    mirror the inserter has extended meanwhile (the new row would become a dead placeholder for good).  LOOKUP
    reports a content version, the UPLOAD carries it, and the server refuses the snapshot when the mirror has
    changed in between — whatever generation names are in play."""
    dim, m, n, efs = 24, 5, 300, 24
    port, X = port_index(n, dim, m, 24, efs, pg.DIST_L2, seed=71)
    meta = pg.make_meta(dim, m, 24, efs, pg.DIST_L2)
    hdr = struct.Struct("<IHhIIQQQQ")
    key = 77

    def call(op, aux=0, gen=0, a0=0, a1=0, payload=b"", fd=None, key=key):
        s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        s.settimeout(10)
        s.connect(srv.socket_path)
        msg = hdr.pack(0x31534748, op, 0, len(payload), aux, key, gen, a0, a1) + payload
        if fd is None:
            s.sendall(msg)
        else:
            socket.send_fds(s, [msg], [fd])
            os.close(fd)
        buf = b""
        while len(buf) < 48:
            buf += s.recv(65536)
        r = hdr.unpack(buf[:48])
        while len(buf) < 48 + r[3]:
            buf += s.recv(65536)
        s.close()
        return r, buf[48:48 + r[3]]

    def version(key=key):
        r, p = call(2, key=key)
        return struct.unpack("<Q", p)[0], r

    def upload(gen, guard, count=n, key=key):
        img = port.raw()[:count * meta.size_data_per_element].tobytes()
        return call(3, gen=gen, a0=count, a1=guard, payload=bytes(meta), fd=_sealed_memfd(img), key=key)[0]

    v0, r = version()
    assert v0 == 0 and r[8] == 0                                    # absent
    assert upload(5, v0 + 1)[2] == 0                                # guarded by "absent": accepted
    v1, r = version()
    assert v1 != 0 and r[6] == 5 and r[7] == n
    assert upload(6, 0 + 1)[2] == HGS_ERR_STALE                     # "I saw no mirror" — but there is one now
    # another backend changes the mirror (here: a vacuum flag) between this backend's LOOKUP and its UPLOAD
    c = RemoteClient(srv.socket_path)
    c.set_deleted(key, 3, True)
    r = upload(6, v1 + 1)
    assert r[2] == HGS_ERR_STALE and r[6] == 5                      # refused; the answer names the current generation
    assert c.lookup(key) == (True, 5, n)
    v2, _ = version()
    assert v2 != v1
    assert upload(6, v2 + 1)[2] == 0                                # a walk that started after the change is taken
    assert c.lookup(key) == (True, 6, n)
    assert upload(7, 0)[2] == 0                                     # unguarded (a host that owns the key outright)
    # "absent" is versioned too.  A scanner LOOKUPs a key that has no mirror and walks the index while
    # a VACUUM flips flags in place; the VACUUM ends with a DROP (of a key that may have no mirror: NOKEY); the scanner's
    # guarded UPLOAD of its pre-VACUUM snapshot must be refused although the key is absent before AND after.
    k2 = 78
    va, r = version(k2)
    assert r[8] == 0                                                # absent
    assert call(6, key=k2)[0][2] == HGS_ERR_NOKEY                    # DROP of a key without a mirror ...
    vb, r = version(k2)
    assert r[8] == 0 and vb != va                                   # ... still absent, but not the same "absent"
    assert upload(5, va + 1, key=k2)[2] == HGS_ERR_STALE             # the snapshot walked before the DROP is refused
    assert upload(5, vb + 1, key=k2)[2] == 0                         # one walked after it is taken
    # the same when the mirror existed at neither end but in between (uploaded and dropped by others)
    k3 = 79
    vc, _ = version(k3)
    assert upload(9, 0, key=k3)[2] == 0 and call(6, key=k3)[0][2] == 0
    assert upload(9, vc + 1, key=k3)[2] == HGS_ERR_STALE
    # an insert lands on a mirror that already holds the new row as the walk's zero placeholder (the host had
    # stored it, still unlinked, when the snapshot was taken): BIND gives it its row, then links it
    new = gmm(1, dim, k=20, seed=72)[0]
    ph = np.zeros(meta.size_data_per_element, np.uint8)
    ph[meta.offset_label:meta.offset_label + 8] = np.frombuffer(struct.pack("<Q", 1 << 48), np.uint8)   # dead placeholder
    img = port.raw().tobytes() + ph.tobytes()
    assert call(3, gen=8, a0=n + 1, payload=bytes(meta), fd=_sealed_memfd(img))[0][2] == 0
    r, _ = call(9, aux=n, gen=8, a0=4242, payload=new.astype(np.float32).tobytes())
    assert r[2] == 0
    port.add(new[None, :], np.array([4242], np.uint64))
    for q in list(X[:5]) + [new]:
        assert (c.search(key, q, efs)[0] == port.search(q, efs)[0]).all()
    assert 4242 in c.search(key, new, efs)[0].tolist()
    c.close()


def test_row_shards_behind_one_front(double_bin):
    """SURVEY.md 8e mode 2 for process-per-connection hosts (round 6; include/hnsw_gpu_server.h, HGS_OP_SHARD_*): two servers hold one row
    shard each of an index under the SAME key; the one started with --shard-peers is the front backends talk to — it searches its own
    shard, has the peer search the same queries into the exchange buffer it shares (hnsw_gpu_shared_alloc / _open: POSIX shared memory in
    the engine double, an IPC-mapped device allocation in the product), merges by (distance, label) and answers.  Parity as the layout
    defines it: the oracle per shard + a CPU merge, id lists and distance bits, from several backends at once (batches form)."""
    dim, m, efs, n0, n1 = 24, 6, 32, 900, 700
    X = gmm(n0 + n1, dim, k=20, seed=77)
    shards = []
    for lo, hi in ((0, n0), (n0, n0 + n1)):
        p = oracle.PortIndex(dim, m, 32, efs, pg.DIST_L2)
        p.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64) + 10_000)            # labels unique across the shards
        shards.append(p)
    meta = pg.make_meta(dim, m, 32, efs, pg.DIST_L2)
    key = 99
    with ServerProcess(binary=double_bin, lanes=0) as peer:
        with ServerProcess(binary=double_bin, lanes=0, shard_peers=[peer.socket_path]) as front:
            RemoteClient(front.socket_path).upload(meta, key, 1, shards[0].raw(), n0)
            RemoteClient(peer.socket_path).upload(meta, key, 1, shards[1].raw(), n1)

            def want(q, ef):
                both = [s.search(q, ef)[:2] for s in shards]
                lab = np.concatenate([b[0] for b in both]); dst = np.concatenate([b[1] for b in both])
                order = np.lexsort((lab, dst))[:ef]
                return lab[order], dst[order]

            Q = gmm(120, dim, k=20, seed=77, stream=1)
            errors = []

            def backend(t):
                try:
                    c = RemoteClient(front.socket_path)
                    for i in range(t, len(Q), 6):
                        for ef in (efs, 5):
                            lab, dst = c.search(key, Q[i], ef)
                            wl, wd = want(Q[i], ef)
                            if not ((lab == wl).all() and (bits(dst) == bits(wd)).all()):
                                errors.append((i, ef))
                except Exception as e:                             # noqa: BLE001
                    errors.append(repr(e))
            th = [threading.Thread(target=backend, args=(t,)) for t in range(6)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errors, errors[:5]
            sf, sp = RemoteClient(front.socket_path).stats(), RemoteClient(peer.socket_path).stats()
            assert sf["searches"] == 240 and sp["searches"] == 240 and sf["search_errors"] == 0 and sp["search_errors"] == 0
            # a vacuum flag set on the PEER's shard is honoured in the merged answer (the peer filters its own list)
            l0 = RemoteClient(front.socket_path).search(key, Q[0], efs)[0]
            victim = next(int(x) for x in l0 if x >= 10_000 + n0)
            RemoteClient(peer.socket_path).set_deleted(key, victim - 10_000 - n0, True)
            shards[1].set_deleted(victim - 10_000 - n0, True)
            lab, dst = RemoteClient(front.socket_path).search(key, Q[0], efs)
            assert victim not in lab.tolist() and (lab == want(Q[0], efs)[0]).all()
        # the peer outlives the front; a front whose peer is gone answers with an error, not with half an answer
    with ServerProcess(binary=double_bin, lanes=0, shard_peers=["/nonexistent/hgs-peer"]) as lonely:
        c = RemoteClient(lonely.socket_path)
        c.upload(meta, key, 1, shards[0].raw(), n0)
        with pytest.raises(RemoteError):
            c.search(key, X[0], efs)
    with pytest.raises(RuntimeError):                              # a front takes blocking batches only
        ServerProcess(binary=double_bin, lanes=2, shard_peers=["/x"]).start()
