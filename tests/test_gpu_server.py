"""hnsw_gpu_server on the device: the shipped binary (linked against libhnsw_gpu.so) behind the
client library.  Every answer that comes back through the socket is compared with the oracle —
bit-exact labels and distances, as for the in-process path."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from pg_embedding_amd.server import RemoteClient, ServerProcess
import server_util as SU
from test_server_cpu import bits, port_index, run_clients

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=[2, 0, "stream"], ids=["streamed", "blocking", "resident"])
def srv(request):
    """The dispatcher forms: streamed completion on lanes (kernel writes results + per-query flags straight into pinned host memory,
    answers leave as their walks end), one blocking launch at a time, and --stream 1: ONE resident launch per (mirror, efsearch) fed
    through a ring in pinned memory (include/hnsw_gpu.h, "Streams"; a small ring here, so that every slot is reused many times)."""
    if request.param == "stream":
        with ServerProcess(stream=True, ring=256, dispatchers=2) as s:
            yield s
    else:
        with ServerProcess(lanes=request.param) as s:
            yield s


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
def test_remote_search_is_bit_exact(srv, func):
    dim, m, n, efs = 96, 8, 4000, 64
    port, X = port_index(n, dim, m, 40, efs, func, seed=70 + func)
    meta = pg.make_meta(dim, m, 40, efs, func)
    c = RemoteClient(srv.socket_path)
    key = 100 + func
    c.upload(meta, key, 1, port.raw(), n)
    Q = gmm(40, dim, k=20, seed=70 + func, stream=1)
    for q in Q:
        for ef in (efs, 10, 300):
            lab, dst = c.search(key, q, ef)
            wl, wd = port.search(q, ef)[:2]
            assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
    c.set_deleted(key, 17, True)
    port.set_deleted(17, True)
    assert (c.search(key, X[17], efs)[0] == port.search(X[17], efs)[0]).all()
    from test_gpu_build import live_image          # link slots past `count` are not mirrored
    assert (live_image(c.export(key, n * meta.size_data_per_element), meta, n) == live_image(port.raw(), meta, n)).all()
    c.drop(key)
    c.close()


def test_many_backends_one_device(srv, tmp_path):
    """96 backend processes, one query per hnsw_search call: all answers equal the oracle's, and the
    server served them in batches on two streams."""
    dim, m, n, efs = 128, 8, 20000, 64
    port, X = port_index(n, dim, m, 48, efs, pg.DIST_L2, seed=81)
    meta = pg.make_meta(dim, m, 48, efs, pg.DIST_L2)
    Q = gmm(3840, dim, k=20, seed=81, stream=1)
    c = RemoteClient(srv.socket_path)
    before = c.stats()
    c.upload(meta, 7, 2, port.raw(), n)
    info, labels, counts = run_clients(srv.socket_path, 7, 2, dim, m, 48, efs, pg.DIST_L2, Q, 96, tmp_path)
    want = port.search_many(Q, efs, nthreads=8)
    assert (counts == want["counts"]).all()
    for q in range(len(Q)):
        k = int(counts[q])
        assert (labels[q, :k] == want["labels"][q, :k]).all()
    st = c.stats()
    assert st["searches"] - before["searches"] == len(Q) and st["search_errors"] == 0
    assert st["max_batch"] > 4 or "--stream" in srv.args      # (a resident launch forms no batches)
    print("many backends:", info, {k: st[k] - before[k] for k in ("searches", "batches")}, "max batch", st["max_batch"])
    c.drop(7)
    c.close()


def test_dropin_symbols_through_the_server(srv):
    """C host linked against libembedding_gpuc.so, device behind the server: inserts through
    hnsw_bind_point, searches through hnsw_search; same output as linked against the reference."""
    exe = SU.build_c_client("dropin_demo")
    ref = SU.build_c_reference("dropin_demo")
    args = ["400", "24", "4", "16", "12", "15"]
    env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=srv.socket_path)
    attached = subprocess.run([exe] + args + ["555"], capture_output=True, text=True, env=env, check=True).stdout
    plain = subprocess.run([exe] + ["150"] + args[1:], capture_output=True, text=True, env=env, check=True).stdout
    assert attached.count("\n") == 15 and plain.count("\n") == 15
    if ref:
        assert attached == subprocess.run([ref] + args, capture_output=True, text=True, check=True).stdout
        assert plain == subprocess.run([ref] + ["150"] + args[1:], capture_output=True, text=True, check=True).stdout
    c = RemoteClient(srv.socket_path)
    assert c.lookup(555) == (True, 1, 400)
    c.drop(555)
    c.close()


def test_server_side_bulk_build_then_search(srv):
    """UPLOAD zero-linked rows + LINK in the default batched mode (CREATE INDEX offload), EXPORT the
    graph to the host: the oracle searching those bytes returns what the server returns."""
    dim, m, n, efs = 64, 8, 6000, 48
    X = gmm(n, dim, k=30, seed=91)
    meta = pg.make_meta(dim, m, 64, efs, pg.DIST_L2)
    esz = meta.size_data_per_element
    img = np.zeros((n, esz), np.uint8)
    img[:, meta.offset_data:meta.offset_label] = X.view(np.uint8).reshape(n, dim * 4)
    img[:, meta.offset_label:] = (np.arange(n, dtype=np.uint64) + 1).view(np.uint8).reshape(n, 8)
    c = RemoteClient(srv.socket_path)
    c.upload(meta, 31, 1, img.reshape(-1), n)
    c.link(31, 0, n, 0)
    graph = c.export(31, n * esz)
    port = oracle.PortIndex(dim, m, 64, efs, pg.DIST_L2)
    port.load_raw(graph, n)
    hits = 0
    for q in gmm(50, dim, k=30, seed=91, stream=1):
        lab, dst = c.search(31, q, efs)
        wl, wd = port.search(q, efs)[:2]
        assert (lab == wl).all() and (bits(dst) == bits(wd)).all()
        exact = np.argsort(((X - q) ** 2).sum(1), kind="stable")[:10] + 1
        hits += len(set(exact.tolist()) & set(lab[:10].tolist()))
    assert hits / 500 > 0.9                               # the batched device build is a good graph
    c.drop(31)
    c.close()


def test_row_shards_behind_one_front_on_the_device():
    """Two product servers on this device, one row shard each under the same key; the one started with --shard-peers is the front
    (include/hnsw_gpu_server.h, HGS_OP_SHARD_*): its own shard + the peer's lists — written by the PEER PROCESS's kernel into the front's
    exchange buffer through the IPC mapping (hnsw_gpu_shared_alloc / _open; across two GPUs the same stores cross xGMI) — merged by
    (distance, label) on the front's device.  Parity as the layout defines it: oracle per shard + CPU merge, ids and distance bits,
    several backends at once; and what a sharded batch costs next to an unsharded one of the same size."""
    import threading
    import time
    dim, m, efs, n0, n1 = 128, 8, 64, 6000, 5000
    X = gmm(n0 + n1, dim, k=30, seed=91)
    shards = []
    for lo, hi in ((0, n0), (n0, n0 + n1)):
        p = oracle.PortIndex(dim, m, 40, efs, pg.DIST_L2)
        p.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64) + 10_000)
        shards.append(p)
    meta = pg.make_meta(dim, m, 40, efs, pg.DIST_L2)
    key = 7

    def want(q, ef):
        both = [s.search(q, ef)[:2] for s in shards]
        lab = np.concatenate([b[0] for b in both]); dst = np.concatenate([b[1] for b in both])
        order = np.lexsort((lab, dst))[:ef]
        return lab[order], dst[order]

    with ServerProcess(lanes=0, dispatchers=1) as peer:
        with ServerProcess(lanes=0, dispatchers=1, shard_peers=[peer.socket_path]) as front:
            RemoteClient(front.socket_path).upload(meta, key, 1, shards[0].raw(), n0)
            RemoteClient(peer.socket_path).upload(meta, key, 1, shards[1].raw(), n1)
            Q = gmm(160, dim, k=30, seed=91, stream=1)
            errors = []

            def backend(t):
                try:
                    c = RemoteClient(front.socket_path)
                    for i in range(t, len(Q), 8):
                        for ef in (efs, 10):
                            lab, dst = c.search(key, Q[i], ef)
                            wl, wd = want(Q[i], ef)
                            if not ((lab == wl).all() and (bits(dst) == bits(wd)).all()):
                                errors.append((i, ef))
                except Exception as e:                             # noqa: BLE001
                    errors.append(repr(e))
            th = [threading.Thread(target=backend, args=(t,)) for t in range(8)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert not errors, errors[:5]
            sf, sp = RemoteClient(front.socket_path).stats(), RemoteClient(peer.socket_path).stats()
            assert sf["searches"] == 320 and sp["searches"] == 320 and sf["search_errors"] == 0 and sp["search_errors"] == 0
            # one backend alone: round trip of a sharded search (front + peer + merge) against the peer's own unsharded search
            c = RemoteClient(front.socket_path)
            t0 = time.perf_counter()
            for q in Q[:100]:
                c.search(key, q, efs)
            sharded_ms = (time.perf_counter() - t0) * 10.0
            c2 = RemoteClient(peer.socket_path)
            t0 = time.perf_counter()
            for q in Q[:100]:
                c2.search(key, q, efs)
            plain_ms = (time.perf_counter() - t0) * 10.0
            print(f"\n[row shards behind a front, two server processes on one device] one backend: {sharded_ms:.3f} ms per sharded search "
                  f"(front shard + peer shard through the shared buffer + merge) against {plain_ms:.3f} ms for an unsharded search of one shard")
