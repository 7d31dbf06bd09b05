"""bench.py's launch path on CPU: `python bench.py --gpus N` must start N ranks itself (it used to
benchmark one GPU silently), and under the driver's own `torch.distributed.run` command it must join the ranks
it is given.  PGEMB_BENCH_SELFTEST=1 swaps the device work for one gloo collective; the launcher, the argument
passing and the rendezvous on 127.0.0.1 are the product's."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out          # ONE JSON line, from rank 0 only
    return json.loads(lines[0])


def _env():
    e = dict(os.environ)
    e["PGEMB_BENCH_SELFTEST"] = "1"
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID"):
        e.pop(k, None)
    return e


@pytest.mark.parametrize("mode", ["replicas", "sharded"])
def test_plain_invocation_launches_the_ranks_itself(mode):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "0", "--mode", mode],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_joined"] == 2 and j["self_launched"] is True and j["mode"] == mode
    if mode == "sharded":
        # the row-sharded layout's exchange step ran for real over gloo (pg_embedding_amd.sharded.ShardedIndex): ONE collective per
        # search, one set of buffers for all searches, the merged result is the global (dist, label) order on both ranks, and the
        # line carries the per-rank step breakdown the device run reports
        assert j["exchange"]["collectives_per_step"] == 1 and j["buffers_allocated"] == 1
        assert j["merged_equals_the_global_order_on_every_rank"] is True
        assert len(j["step_breakdown_ms_per_rank"]) == 2 and all(set(r) == {"local_search_ms", "exchange_ms", "merge_ms"} for r in j["step_breakdown_ms_per_rank"])


def test_driver_style_launch_is_joined_not_relaunched():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1"],
                       env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["ranks_joined"] == 2 and j["self_launched"] is False


def test_single_process_default_does_not_launch_anything():
    r = subprocess.run([sys.executable, BENCH], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and j["backend"] is None
