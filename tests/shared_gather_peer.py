"""The OTHER process of tests/test_gpu_sharded.py::test_shards_in_two_processes_meet_in_one_shared_buffer: owns shard `r` of the
row-sharded table, maps the merging process's exchange buffer from its IPC handle, searches its shard straight into list r of that
buffer, and says so on stdout once its stream has drained.   usage: shared_gather_peer.py <handle hex> <r> <nshards> <n> <dim> <m> <efc> <ef> <nq> <seed>"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import oracle                                               # noqa: E402
import pg_embedding_amd as pg                               # noqa: E402
from pg_embedding_amd._lib import check, gpu_lib            # noqa: E402
from pg_embedding_amd.datasets import gmm                   # noqa: E402
from pg_embedding_amd.sharded import shard_range            # noqa: E402

handle = bytes.fromhex(sys.argv[1])
r, nshards, n, dim, m, efc, ef, nq, seed = (int(x) for x in sys.argv[2:11])
L = gpu_lib()
X = gmm(n, dim, k=40, seed=seed)
Q = gmm(nq, dim, k=40, seed=seed, stream=1)
lo, hi = shard_range(n, nshards, r)
port = oracle.PortIndex(dim, m, efc, ef, pg.DIST_L2)
port.add(X[lo:hi], np.arange(lo, hi, dtype=np.uint64))
ix = pg.GpuIndex.from_flat(pg.make_meta(dim, m, efc, ef, pg.DIST_L2), port.raw(), hi - lo)
base = C.c_void_p()
check(L.hnsw_gpu_shared_open(0, handle, C.byref(base)), "hnsw_gpu_shared_open")
lab_bytes, dst_bytes = nshards * nq * ef * 8, nshards * nq * ef * 4
labels = base.value + r * nq * ef * 8
dists = base.value + lab_bytes + r * nq * ef * 4
counts = base.value + lab_bytes + dst_bytes + r * nq * 4
dq = torch.from_numpy(Q).cuda()
check(L.hnsw_gpu_search_batch_dev(ix._h, dq.data_ptr(), nq, ef, labels, dists, counts, None, None), "hnsw_gpu_search_batch_dev")
torch.cuda.synchronize()
print("STORED", flush=True)
sys.stdin.readline()                                        # the merging process says when it has merged; the mapping must live until then
check(L.hnsw_gpu_shared_close(0, base), "hnsw_gpu_shared_close")
ix.close()
print("CLOSED", flush=True)
