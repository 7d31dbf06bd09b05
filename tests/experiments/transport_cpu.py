"""CPU time per search of the two transports between a backend and hnsw_gpu_server — socket (sendmsg / epoll / recvmsg on both sides) and the
shared-memory mailbox (HGS_OP_SHM: post without a system call, futex wake to answer) — measured WITHOUT a device: the server's own source
over the engine double with a toy index (a search costs the "device" a few microseconds), P single-threaded C backends calling hnsw_search().
user + system CPU seconds of the server process and of all backends per million searches.
    python tests/experiments/transport_cpu.py [backends] [queries-per-backend-round] [rounds]
"""
import json
import os
import resource
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                         # noqa: E402
import oracle                                              # noqa: E402
import pg_embedding_amd as pg                              # noqa: E402
import server_util as SU                                   # noqa: E402
from pg_embedding_amd.datasets import gmm                  # noqa: E402
from pg_embedding_amd.server import RemoteClient, ServerProcess   # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NQ = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dim, m, n, efs = 8, 4, 64, 4
X = gmm(n, dim, k=4, seed=1)
port = oracle.PortIndex(dim, m, 8, efs, pg.DIST_L2)
port.add(X, np.arange(n, dtype=np.uint64) + 500)
Q = gmm(NQ, dim, k=4, seed=2)
exe = SU.build_c_client("server_clients")
binary = SU.build_double_server()


def proc_cpu(pid):
    f = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()
    tck = os.sysconf("SC_CLK_TCK")
    return (int(f[11]) + int(f[12])) / tck                 # utime + stime of the process (all threads)


for label, pollers, shm in (("socket", None, "0"), ("mailbox", 2, "1"), ("socket", None, "0"), ("mailbox", 2, "1")):
    for stream in (False, True):
        with ServerProcess(binary=binary, lanes=2, dispatchers=2, readers=4, stream=stream, ring=1024, shm_pollers=pollers) as s:
            c = RemoteClient(s.socket_path)
            c.upload(pg.make_meta(dim, m, 8, efs, pg.DIST_L2), 1, 1, port.raw(), n)
            with tempfile.TemporaryDirectory() as td:
                qf, of = os.path.join(td, "q.f32"), os.path.join(td, "out.u64")
                np.ascontiguousarray(Q, np.float32).tofile(qf)
                env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=s.socket_path, PG_EMBEDDING_GPU_SHM=shm)
                s0 = proc_cpu(s.proc.pid)
                r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
                t0 = time.time()
                r = subprocess.run([exe, "1", "1", str(dim), str(m), "8", str(efs), str(pg.DIST_L2), qf, str(NQ), str(P), of, str(ROUNDS)],
                                   capture_output=True, text=True, env=env, timeout=900)
                wall = time.time() - t0
                assert r.returncode == 0, r.stderr[-500:]
                r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
                s1 = proc_cpu(s.proc.pid)
            st = c.stats()
            c.close()
        nsearch = st["searches"]
        cli = (r1.ru_utime - r0.ru_utime) + (r1.ru_stime - r0.ru_stime)
        print(json.dumps({"transport": label, "server": "stream" if stream else "lanes", "backends": P, "searches": nsearch, "through_mailboxes": st["shm_searches"],
                          "wall_s": round(wall, 2), "searches_per_s": round(nsearch / wall),
                          "server_cpu_us_per_search": round((s1 - s0) / nsearch * 1e6, 2),
                          "backends_cpu_us_per_search": round(cli / nsearch * 1e6, 2)}), flush=True)
