"""Soak of hnsw_gpu_server over the engine double (plain / ThreadSanitizer / AddressSanitizer builds): every dispatcher form and transport in turn, twelve
client threads with two beams whose connections come and go, a writer toggling a delete flag; every answer must equal the oracle's for one of the two
states, the server must end cleanly and without a sanitizer report.  Not part of the test tiers (minutes):  python tests/experiments/server_soak.py plain|tsan|asan <iterations>"""
import os, sys, time, threading, subprocess, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PG_EMBEDDING_GPU_SHM"] = "1"
import server_util as SU
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm
from pg_embedding_amd.server import RemoteClient, ServerProcess
import oracle
kind = sys.argv[1]
b = {"asan": SU.build_double_server_asan, "tsan": SU.build_double_server_tsan, "plain": SU.build_double_server}[kind]()
iters = int(sys.argv[2])
dim, m, n, efs = 24, 4, 900, 16
X = gmm(n, dim, k=20, seed=5)
port = oracle.PortIndex(dim, m, 16, efs, pg.DIST_L2)
port.add(X, np.arange(n, dtype=np.uint64) + 500)
Q = gmm(40, dim, k=20, seed=5, stream=1)
want = {ef: [port.search(q, ef)[0] for q in Q] for ef in (8, efs)}
port.set_deleted(3, True)
want_d = {ef: [port.search(q, ef)[0] for q in Q] for ef in (8, efs)}
port.set_deleted(3, False)
import glob
bad_total = 0
for it in range(iters):
    stream, pollers = [(False, None), (True, None), (True, 2), (False, 2)][it % 4]
    log = f"/tmp/soak_{kind}_{it}"
    env = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:exitcode=77", "TSAN_OPTIONS": f"log_path={log} exitcode=0", "HGS_DOUBLE_SLEEP_US": str([0, 300, 1500][it % 3])}
    s = ServerProcess(binary=b, lanes=2, stream=stream, ring=256, shm_pollers=pollers, env=env).start()
    errs = []
    try:
        c0 = RemoteClient(s.socket_path)
        c0.upload(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), 5, 1, port.raw(), n)
        def worker(t):
            try:
                for rep in range(3):
                    c = RemoteClient(s.socket_path)
                    ef = efs if t % 2 else 8
                    for i, q in enumerate(Q):
                        lab = c.search(5, q, ef)[0]
                        if not (np.array_equal(lab, want[ef][i]) or np.array_equal(lab, want_d[ef][i])): errs.append(("wrong", t, i))
                    c.close()
            except Exception as ex:
                errs.append(repr(ex))
        th = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
        [t.start() for t in th]
        for i in range(40): c0.set_deleted(5, 3, i % 2 == 0)
        [t.join() for t in th]
        st = c0.stats(); c0.close()
    finally:
        rc = s.stop()
    reports = sum(open(f).read().count("WARNING: ThreadSanitizer") for f in glob.glob(log + "*"))
    ok = not errs and rc == 0 and st["search_errors"] == 0 and reports == 0
    bad_total += 0 if ok else 1
    print(it, "stream" if stream else "lanes", "mailbox" if pollers else "socket", "ok" if ok else ("FAIL", errs[:2], rc, st["search_errors"], reports), flush=True)
print("BAD", bad_total)
