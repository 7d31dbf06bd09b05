"""Is the headline index the reference's workload?  (hnswalg.cpp:279-291)

usage: scripts/exp_serial_vs_batched.py [--timeout S]
bench.py's `serial_vs_batched_build` leg on its own: the headline table (1M x 768, L2, m=16, efconstruction=200) as the reference
itself builds it (oracle/_ref's serial graph, tests/experiments/make_ref_serial_graph.py; the link words travel in oracle/_ref/) and as the
batched device builder builds it, same rows, same 40 000 queries — E_q, H_q, recall@10, mean degree, q/s side by side.  Without the
file: `--serial-rows` rows built twice on the device (serial = max_batch 1 = the oracle's bytes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pg_embedding_amd import watchdog; watchdog.arm()
import torch
import pg_embedding_amd as pg
import bench
args = bench.parse()
dev = torch.device("cuda", 0)
res = bench.serial_vs_batched(args, dev, 0, pg.DIST_L2, min(args.serial_rows, args.n))
print(json.dumps(res, indent=1))
for k in ("serial", "batched"):
    r = res[k]
    print(f"{k:8s} E_q {r['evals_per_query']:8.2f}  H_q {r['hops_per_query']:7.2f}  recall@10 {r['recall_at_10']:.4f}  mean degree {r['mean_degree']:.2f}  "
          f"full lists {r['full_lists']:.3f}  B_q {r['alg_bytes_per_query'] / 1e6:.3f} MB  {r['queries_per_s'] / 1e6:.3f} M q/s ({r['kernel_ms_per_launch']:.2f} ms)")
print("serial graph built by:", res["serial_graph_built_by"])
print("batched - serial, relative:", {k: round(v, 4) for k, v in res["batched_minus_serial_relative"].items()}, "within 2 %:", res["within_2_percent"])
