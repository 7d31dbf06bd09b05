"""BASELINE.json's full-size configuration (1M x 768, L2, efsearch=128) through size-independent
properties, plus a sampled bit-exact comparison with the CPU oracle on the exported graph bytes."""
import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm_torch, recall_at_k
from util import bits

pytestmark = pytest.mark.gpu

N, DIM, M, EFC, EF = 1_000_000, 768, 16, 200, 128


@pytest.fixture(scope="module")
def big():
    import torch
    dev = torch.device("cuda", 0)
    X = gmm_torch(N, DIM, device=dev)
    meta = pg.make_meta(DIM, M, EFC, EF, pg.DIST_L2)
    ix = pg.GpuIndex.empty(meta, N)
    ix.append_torch(X)
    ix.link(0, N)
    torch.cuda.synchronize()
    Q = gmm_torch(4000, DIM, stream=1, device=dev)
    yield ix, X, Q
    ix.close()


def test_results_are_sorted_full_and_idempotent(big):
    import torch
    ix, X, Q = big
    a = ix.search_torch(Q, EF, stats=True)
    torch.cuda.synchronize()
    la, da, ca = a["labels"].clone(), a["dists"].clone(), a["counts"].clone()
    assert (ca == EF).all()
    assert (da[:, 1:] >= da[:, :-1]).all()                          # ascending distances
    assert int(la.min()) >= 0 and int(la.max()) < N
    srt = torch.sort(la, dim=1).values
    assert (srt[:, 1:] != srt[:, :-1]).all()                        # no label twice in one result
    b = ix.search_torch(Q, EF)
    torch.cuda.synchronize()
    assert (b["labels"] == la).all() and (b["dists"].view(torch.int32) == da.view(torch.int32)).all()
    # a result's distance is the canonical distance of that row (checksum of checksums over a sample)
    q = 17
    rows = X[la[q]].cpu().numpy()
    d = pg.dist_batch(pg.DIST_L2, Q[q].cpu().numpy(), rows)
    assert (bits(d) == bits(da[q].cpu().numpy())).all()
    st = a["stats"].cpu().numpy()
    assert st[:, 0].min() >= EF and st[:, 1].min() >= 1


def test_recall_gate_of_the_metric(big):
    import torch
    ix, X, Q = big
    truth, tdist = ix.bruteforce_torch(Q[:500].contiguous(), 10, mfma=True)
    out = ix.search_torch(Q[:500].contiguous(), EF)
    torch.cuda.synchronize()
    rec = recall_at_k(out["labels"].cpu().numpy(), truth.cpu().numpy(), 10)
    assert rec >= 0.95, rec
    # the exhaustive scorer itself: canonical scan and MFMA filter agree at full size
    t2, d2 = ix.bruteforce_torch(Q[:64].contiguous(), 10)
    assert (t2 == truth[:64]).all() and (d2.view(torch.int32) == tdist[:64].view(torch.int32)).all()


def test_cpu_oracle_agrees_on_the_exported_graph(big):
    """export -> (host image) -> CPU oracle on the same bytes: bit-exact on a sample of queries;
    and export -> import is the identity."""
    import torch
    ix, X, Q = big
    raw = ix.export_flat()
    port = oracle.PortIndex(DIM, M, EFC, EF, pg.DIST_L2, capacity=N)
    port.load_raw(raw, N)
    Qh = Q[:96].cpu().numpy()
    want = port.search_many(Qh, EF, nthreads=16)
    out = ix.search_torch(Q[:96].contiguous(), EF, stats=True)
    torch.cuda.synchronize()
    assert (out["labels"].cpu().numpy().view(np.uint64) == want["labels"]).all()
    assert (bits(out["dists"].cpu().numpy()) == bits(want["dists"])).all()
    st = out["stats"].cpu().numpy().astype(np.uint32)
    assert (st[:, 0] == want["evals"]).all() and (st[:, 1] == want["hops"]).all()
    if oracle.have_ref():
        ref = oracle.RefIndex(DIM, M, EFC, EF, pg.DIST_L2, capacity=N)
        ref.load_raw(raw, N)
        r = ref.search_many(Qh, EF, nthreads=16)
        same = (r["labels"] == want["labels"]).all(axis=1).mean()
        assert same >= 0.95                       # the rest differ only at near-ties (tests/test_gpu_search.py)
        del ref
    again = pg.GpuIndex.from_flat(ix.meta, raw, N)
    assert (again.export_flat() == raw).all()
    again.close()
