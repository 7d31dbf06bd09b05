"""Churn soak of hnsw_gpu_server over the engine double: searches on one mirror (two beams, connections that come and go) while another
client uploads, searches, re-uploads, updates and drops OTHER mirrors and toggles delete flags, under a "device" too small for all of
them (evictions).  Every answer for the steady mirror must equal the oracle's; the churned mirrors may answer or refuse (no key / stale),
never wrongly; the server must end cleanly.    python tests/experiments/server_churn.py plain|tsan|asan <iterations>"""
import glob
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PG_EMBEDDING_GPU_SHM"] = "1"
import numpy as np                                         # noqa: E402
import oracle                                              # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
import server_util as SU                                   # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402
from pg_embedding_amd.server import HGS_ERR_NOKEY, HGS_ERR_STALE, RemoteClient, RemoteError, ServerProcess   # noqa: E402

kind, iters = sys.argv[1], int(sys.argv[2])
binary = {"asan": SU.build_double_server_asan, "tsan": SU.build_double_server_tsan, "plain": SU.build_double_server}[kind]()
dim, m, efs = 24, 4, 16
meta = pg.make_meta(dim, m, 16, efs, pg.DIST_L2)


def index(n, seed):
    X = gmm(n, dim, k=20, seed=seed)
    p = oracle.PortIndex(dim, m, 16, efs, pg.DIST_L2)
    p.add(X, np.arange(n, dtype=np.uint64) + 500)
    return p, X


steady, Xs = index(900, 5)
others = [index(400 + 50 * i, 20 + i) for i in range(4)]
Q = gmm(30, dim, k=20, seed=6)
want = {ef: [steady.search(q, ef)[0] for q in Q] for ef in (8, efs)}
want_o = [[p.search(q, efs)[0] for q in Q] for p, _ in others]
bad_total = 0
for it in range(iters):
    stream, pollers = [(False, None), (True, None), (True, 2), (False, 2)][it % 4]
    if os.environ.get("CHURN_ONLY") == "stream":
        stream, pollers = True, (2 if it % 2 else None)
    log = f"/tmp/churn_{kind}_{it}"
    env = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=77", "TSAN_OPTIONS": f"log_path={log} exitcode=0",
           "HGS_DOUBLE_SLEEP_US": str([0, 400][it % 2]), "HGS_DOUBLE_CAPACITY": "2000"}      # steady 900 + at most two others fit
    s = ServerProcess(binary=binary, lanes=2, stream=stream, ring=256, shm_pollers=pollers, env=env).start()
    errs, stop = [], threading.Event()
    try:
        c0 = RemoteClient(s.socket_path)
        c0.upload(meta, 5, 1, steady.raw(), 900)

        def searcher(t):
            try:
                while not stop.is_set():
                    c = RemoteClient(s.socket_path)
                    ef = efs if t % 2 else 8
                    for i, q in enumerate(Q):
                        try:
                            lab = c.search(5, q, ef)[0]
                        except RemoteError as ex:
                            if ex.code == HGS_ERR_NOKEY:               # the steady mirror was evicted to make room: put it back
                                c.upload(meta, 5, 1, steady.raw(), 900)
                                continue
                            raise
                        if not np.array_equal(lab, want[ef][i]):
                            errs.append(("wrong", t, i))
                    c.close()
            except Exception as ex:            # noqa: BLE001
                errs.append(repr(ex))

        def churner():
            try:
                c = RemoteClient(s.socket_path)
                gen = 1
                for rnd in range(12):
                    for k, (p, X) in enumerate(others):
                        key = 100 + k
                        try:
                            c.upload(meta, key, gen, p.raw(), p.count)
                            for i, q in enumerate(Q[:6]):
                                lab = c.search(key, q, efs)[0]
                                if not np.array_equal(lab, want_o[k][i]):
                                    errs.append(("wrong on churned mirror", k, i))
                            if rnd % 3 == 0:
                                c.update(meta, key, gen, gen + 1, p.raw()[:10 * meta.size_data_per_element], 0, 10)
                                c.search(key, Q[0], efs, gen=gen + 1)
                            if rnd % 2:
                                c.drop(key)
                        except RemoteError as ex:
                            if ex.code not in (HGS_ERR_NOKEY, HGS_ERR_STALE, -3):       # (-3: the double's "out of device memory" when nothing can be evicted)
                                raise
                    gen += 2
                c.close()
            except Exception as ex:            # noqa: BLE001
                errs.append(repr(ex))

        th = [threading.Thread(target=searcher, args=(t,)) for t in range(8)]
        ch = threading.Thread(target=churner)
        [t.start() for t in th]
        ch.start()
        ch.join()
        stop.set()
        [t.join() for t in th]
        st = c0.stats()
        c0.close()
    finally:
        rc = s.stop()
    reports = sum(open(f).read().count("WARNING: ThreadSanitizer") for f in glob.glob(log + "*"))
    ok = not errs and rc == 0 and reports == 0
    bad_total += 0 if ok else 1
    print(it, "stream" if stream else "lanes", "mailbox" if pollers else "socket", "ok" if ok else ("FAIL", errs[:3], rc, reports),
          {k: st[k] for k in ("searches", "search_errors", "uploads", "updates", "evictions")}, flush=True)
print("BAD", bad_total)
sys.exit(1 if bad_total else 0)
