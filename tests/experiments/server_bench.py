"""Throughput of the drop-in symbol hnsw_search() when many single-threaded backends share one
hnsw_gpu_server (the Postgres deployment shape, include/hnsw_gpu_server.h).

  * rows are generated on the host, uploaded zero-linked and linked on the server (LINK, batched
    device build) — the CREATE INDEX offload;
  * for each process count P, tests/dropin_c/server_clients.c forks P backends that each call
    hnsw_search() one query at a time; the server coalesces what is waiting into batch launches;
  * a sample of the returned arrays is compared with the reference's own code (oracle/_ref, or the
    C restatement) searching the exported graph bytes.

Usage: python tests/experiments/server_bench.py [--rows 1000000 --dims 768 --m 16 --efc 200 --efs 128]
                                      [--procs 1,16,64,256,1024] [--queries 20480] [--dispatchers 2]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round


def thread_cpu(pid):
    """{tid: (thread name, user + system CPU seconds)} of a process, from /proc"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir(f"/proc/{pid}/task"):
            with open(f"/proc/{pid}/task/{tid}/stat") as f:
                st = f.read()
            name = st[st.index("(") + 1:st.rindex(")")]
            f2 = st[st.rindex(")") + 2:].split()
            out[tid] = (name, (int(f2[11]) + int(f2[12])) / tick)
    except OSError:
        pass
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--dims", type=int, default=768)
    ap.add_argument("--m", type=int, default=16)
    ap.add_argument("--efc", type=int, default=200)
    ap.add_argument("--efs", type=int, default=128)
    ap.add_argument("--procs", default="1,16,64,256,1024")
    ap.add_argument("--queries", type=int, default=20480)
    ap.add_argument("--target-seconds", type=float, default=4.0)
    ap.add_argument("--dispatchers", type=int, default=2)
    ap.add_argument("--readers", type=int, default=8)
    ap.add_argument("--lanes", type=int, default=3, help="launches in flight per dispatcher (0 = blocking batches)")
    ap.add_argument("--hwq", type=int, default=0, help="GPU_MAX_HW_QUEUES for the server process (0 = leave it alone)")
    ap.add_argument("--binary", default=None, help="server binary (default: the shipped one)")
    ap.add_argument("--check", type=int, default=200, help="queries compared with the CPU reference")
    ap.add_argument("--server-env", default="", help="K=V[,K=V] added to the server's environment (e.g. LD_PRELOAD of a variant library)")
    ap.add_argument("--walkers", default=None, help="server --walkers (auto | 0..8): walking waves per block of a search launch")
    ap.add_argument("--shm-pollers", type=int, default=None, help="server --shm-pollers (0 = no mailboxes: every search on the socket)")
    ap.add_argument("--verbose-server", action="store_true", help="server --verbose: sessions opened / closed with their team geometry on stderr")
    ap.add_argument("--stream", action="store_true", help="server --stream 1: one resident launch fed through a ring instead of launches on lanes")
    ap.add_argument("--configs", default="", help="sweep: comma-separated dispatchers:lanes[:readers[:walkers[:stream]]] — one server per entry over the "
                                                  "same rows (data generated once); the parity check runs for the first entry only")
    a = ap.parse_args()
    if a.configs:
        cfgs = []
        for c in a.configs.split(","):
            f = c.split(":")
            cfgs.append((int(f[0]), int(f[1]), int(f[2]) if len(f) > 2 else a.readers, (f[3] or None) if len(f) > 3 else a.walkers,
                         len(f) > 4 and f[4] == "stream"))
    else:
        cfgs = [(a.dispatchers, a.lanes, a.readers, a.walkers, a.stream)]

    import oracle
    import pg_embedding_amd as pg
    from pg_embedding_amd.datasets import gmm
    from pg_embedding_amd.server import RemoteClient, ServerProcess
    import server_util as SU

    exe = SU.build_c_client("server_clients")
    meta = pg.make_meta(a.dims, a.m, a.efc, a.efs, pg.DIST_L2)
    esz = meta.size_data_per_element
    t = time.time()
    X = gmm(a.rows, a.dims, k=1000, seed=42)
    Q = gmm(a.queries, a.dims, k=1000, seed=42, stream=1)
    img = np.zeros((a.rows, esz), np.uint8)
    img[:, meta.offset_data:meta.offset_label] = X.view(np.uint8).reshape(a.rows, a.dims * 4)
    img[:, meta.offset_label:] = np.arange(a.rows, dtype=np.uint64).view(np.uint8).reshape(a.rows, 8)
    del X
    print(f"# data: {a.rows} x {a.dims} GMM(1000, 0.3) in {time.time() - t:.1f} s", flush=True)

    tmp = tempfile.mkdtemp(prefix="hgs_bench_")
    qf, of = os.path.join(tmp, "q.f32"), os.path.join(tmp, "out.u64")
    Q.tofile(qf)
    key, gen = 1, 1
    table = []
    for ci, (nd, nl, nr, nw, strm) in enumerate(cfgs):
      print(f"## server with {nd} dispatchers x {nl} lanes, {nr} readers, walkers {nw or 'default (auto)'}" + (", STREAM mode" if strm else ""), flush=True)
      srv = ServerProcess(dispatchers=nd, readers=nr, lanes=nl, binary=a.binary, walkers=nw, stream=strm, verbose=a.verbose_server, shm_pollers=a.shm_pollers,
                          env=dict(([("GPU_MAX_HW_QUEUES", str(a.hwq))] if a.hwq else []) +
                                   [tuple(kv.split("=", 1)) for kv in a.server_env.split(",") if "=" in kv]) or None)
      with srv:
          c = RemoteClient(srv.socket_path)
          t = time.time()
          c.upload(meta, key, gen, img.reshape(-1), a.rows)
          t_up = time.time() - t
          t = time.time()
          c.link(key, 0, a.rows, 0)
          t_link = time.time() - t
          print(f"# upload {img.nbytes / 1e9:.2f} GB through a memfd: {t_up:.2f} s; device build (LINK): {t_link:.2f} s", flush=True)
          env = dict(os.environ, PG_EMBEDDING_GPU_SERVER=srv.socket_path)
          rows = []
          labels = None
          for P in [int(x) for x in a.procs.split(",")]:
              nq = max(P, min(a.queries, max(P * 8, 512)))
              # short calibration run, then as many rounds as fill the target time
              args = [exe, str(key), str(gen), str(a.dims), str(a.m), str(a.efc), str(a.efs), "0", qf, str(nq), str(P), of]
              r = subprocess.run(args + ["1"], capture_output=True, text=True, env=env, timeout=600)
              assert r.returncode == 0, r.stderr
              cal = json.loads(r.stdout)
              rounds = int(max(1, min(200, a.target_seconds * cal["qps"] / nq)))
              before = c.stats()
              cpu0 = thread_cpu(srv.proc.pid)
              wall0 = time.time()
              r = subprocess.run(args + [str(rounds)], capture_output=True, text=True, env=env, timeout=600)
              assert r.returncode == 0, r.stderr
              info = json.loads(r.stdout)
              cpu1, wall = thread_cpu(srv.proc.pid), time.time() - wall0
              # how busy the server's threads were over the run (user + system CPU seconds / wall seconds), by thread name
              busy = {}
              for tid, (name, sec) in cpu1.items():
                  d0 = cpu0.get(tid, (name, 0.0))[1]
                  busy.setdefault(name, []).append(round((sec - d0) / max(wall, 1e-9), 2))
              info["server_thread_busy_fraction"] = {k: sorted(v, reverse=True) for k, v in busy.items()}
              st = c.stats()
              d = {k: st[k] - before[k] for k in ("searches", "batches", "batch_ns", "kernel_ns", "queue_ns", "walk_ns", "answer_ns")}
              info.update(mean_batch=d["searches"] / max(1, d["batches"]), batches=d["batches"],
                          ms_per_batch=d["batch_ns"] / 1e6 / max(1, d["batches"]),
                          kernel_ms_per_batch=d["kernel_ns"] / 1e6 / max(1, d["batches"]),
                          latency_ms=1e3 * P / info["qps"],
                          # inside the server, per search: waiting for a lane / launch to completion flag seen / writing the answer
                          queue_ms=d["queue_ns"] / 1e6 / max(1, d["searches"]), walk_ms=d["walk_ns"] / 1e6 / max(1, d["searches"]),
                          answer_ms=d["answer_ns"] / 1e6 / max(1, d["searches"]))
              rows.append(info)
              print(json.dumps(info), flush=True)
              out = np.fromfile(of, np.uint64)
              labels = (nq, out[:nq * a.efs].reshape(nq, a.efs).copy(), out[nq * a.efs:].copy())
          # parity of what the backends received, against the reference's code on the same graph bytes
          graph = c.export(key, a.rows * esz)
          nq, lab, cnt = labels
          ncheck = min(a.check, nq)
          checks = [("C restatement in the device's summation order (oracle/hnsw_port.c)", oracle.PortIndex)]
          if oracle.have_ref() and ci == 0:               # (the sweep's later servers: the bit-exact oracle only)
              checks.append(("reference binary, -Ofast summation order (oracle/_ref)", oracle.RefIndex))
          for kind, cls in checks:
              cpu = cls(a.dims, a.m, a.efc, a.efs, pg.DIST_L2)
              cpu.load_raw(graph, a.rows)
              same = 0
              for q in range(ncheck):
                  w = cpu.search(Q[q], a.efs)
                  w = w[0] if isinstance(w, tuple) else w
                  same += int(cnt[q] == len(w) and (lab[q, :len(w)] == w).all())
              print(f"# parity: {same}/{ncheck} sampled hnsw_search() answers identical to the {kind} on the exported graph", flush=True)
              del cpu
          st = c.stats()
          print("# server totals:", json.dumps({k: st[k] for k in ("connections", "searches", "shm_searches", "batches", "max_batch", "search_errors")}))
          c.close()
          table += [(nd, nl, nr, nw, strm, r) for r in rows]
    print("\n| dispatchers x lanes (readers, walkers) | backends | queries/s | mean batch | ms per batch (host) | kernel ms per batch | round trip ms | in server: queue + walk + answer ms |\n|---|---|---|---|---|---|---|---|")
    for nd, nl, nr, nw, strm, r in table:
        print(f"| {'stream, ' + str(nd) + ' answer threads' if strm else str(nd) + ' x ' + str(nl)} ({nr}, {nw or 'auto'}) | {r['nproc']} | {r['qps']:.0f} | {r['mean_batch']:.1f} | {r['ms_per_batch']:.2f} | {r['kernel_ms_per_batch']:.2f} | {r['latency_ms']:.2f} | {r['queue_ms']:.2f} + {r['walk_ms']:.2f} + {r['answer_ms']:.3f} |")


if __name__ == "__main__":
    main()
