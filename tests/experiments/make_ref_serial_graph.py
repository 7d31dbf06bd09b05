"""The reference's OWN serial graph of the headline table: oracle/_ref — the unmodified hnswalg.cpp + distfunc.c —
inserts the 1 000 000 x 768 rows one by one (hnsw_bind_point, hnswalg.cpp:279-291; m = 16, efconstruction = 200, L2) on one host
core of THIS container (it needs /root/reference's compiled objects and about half an hour; the device box has neither the time nor
the budget), and the link words of every element (count + maxM ids, 132 bytes each) are saved to oracle/_ref/ — git-ignored like the
reference binaries, travelling to the device box like them.  Rows: pg_embedding_amd.datasets.gmm (numpy, seeded): the same bytes on
both boxes.  scripts/exp_ref_graph_vs_batched.py searches that graph on the device beside the batched device build of the same rows.

usage: python tests/experiments/make_ref_serial_graph.py [rows=1000000] [dim=768]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import oracle
from pg_embedding_amd.datasets import gmm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 768
m, efc = 16, 200
out = os.path.join(ROOT, "oracle", "_ref", f"serial_graph_{n}x{dim}_m{m}_efc{efc}_l2.npy")
X = gmm(n, dim, k=1000, sigma=0.3, seed=42)
R = oracle.RefIndex(dim, m, efc, 128, oracle.DIST_L2, n)
t0 = time.time()
step = 20000
for a in range(0, n, step):
    R.add(X[a:a + step], np.arange(a, min(n, a + step), dtype=np.uint64))
    el = time.time() - t0
    print(f"{min(n, a + step):>8d} rows, {el:7.1f} s, {min(n, a + step) / el:7.0f} inserts/s", flush=True)
links = R.raw_view().reshape(n, -1)[:, :(2 * m + 1) * 4].copy().view(np.uint32)
np.save(out, links)
print("saved", out, links.shape, f"mean degree {links[:, 0].mean():.3f}, build {time.time() - t0:.0f} s on one host core")
