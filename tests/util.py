"""Shared helpers for the parity tests."""
import numpy as np

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm

REL_TOL = 1e-5      # north-star tolerance against the reference binary
ABS_FLOOR = 1e-6    # |d| floor: relative error is meaningless for cosine distances near 0


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), ABS_FLOOR)


def build_port(n, dim, m, efc, func, k=50, seed=1, data=None):
    X = gmm(n, dim, k=k, seed=seed) if data is None else data
    p = oracle.PortIndex(dim, m, efc, 64, func)
    p.add(X)
    return p, X


def mirror(port_or_flat, func, device=0, efs=64):
    meta = pg.make_meta(port_or_flat.dim, port_or_flat.m, port_or_flat.efc, efs, func)
    return pg.GpuIndex.from_flat(meta, port_or_flat.raw(), port_or_flat.count, device=device)


def near_tie_mask(dists, rel=REL_TOL):
    """positions i where dists[i] is within `rel` of a neighbour in the sorted list."""
    d = np.asarray(dists, np.float64)
    m = np.zeros(d.shape, bool)
    if d.size > 1:
        close = np.abs(np.diff(d)) <= rel * np.maximum(np.abs(d[1:]), ABS_FLOOR)
        m[1:] |= close
        m[:-1] |= close
    return m


def classify_against_reference(port, Q, ef, ref_labels, nthreads=8):
    """The north-star id contract, proven query by query (oracle/hnsw_port.c, PortStats.div_*).

    `port` walks in the canonical (device) arithmetic and shadows every decision of the walk — stop test
    (hnswalg.cpp:70), accept test (:99), which candidate is popped, which result is evicted, output order —
    with the reference's own hnsw_dist_func.  Asserts
      * a query with no diverging decision has the reference's id array (a theorem: both arithmetics then
        walk identically element for element — so this checks the restatement and the harness);
      * a query whose ids differ has a diverging decision, and the two values that decision compared are
        within REL_TOL (1e-5 relative, the north-star tolerance — not a multiple of it) in the walk's own
        arithmetic, i.e. the mismatch is a near-tie flipped by float summation order and nothing else.
    Returns the port's result dict plus the classification counts (what bench.py reports)."""
    port.shadow_reference_distances(True)
    try:
        got = port.search_many(Q, ef, nthreads=nthreads)
    finally:
        port.shadow_reference_distances(False)
    same = (got["labels"] == ref_labels).all(axis=1)
    dk, dm = got["div_kind"], got["div_margin"]
    unexplained = (~same) & (dk == 0)
    assert not unexplained.any(), f"queries {np.flatnonzero(unexplained)[:8]}: ids differ from the reference without any diverging decision"
    worst = float(dm[~same].max()) if (~same).any() else 0.0
    assert worst <= REL_TOL, f"a mismatching query diverged at a decision with relative gap {worst:.3g} > {REL_TOL}"
    got["classification"] = {
        "queries": int(Q.shape[0]),
        "identical_ids": int(same.sum()),
        "mismatch_count": int((~same).sum()),
        "mismatch_explained_by_near_tie": int(((~same) & (dk != 0)).sum()),
        "mismatch_unexplained": int(unexplained.sum()),
        "largest_gap_at_a_mismatching_decision": worst,
        "queries_with_a_diverging_decision": int((dk != 0).sum()),
        "diverging_decision_kinds": {k: int((dk == i).sum()) for i, k in
                                     ((1, "stop"), (2, "accept"), (3, "pop_order"), (4, "evict"), (5, "output_order"))},
    }
    return got


def evals_from_pops(raw, meta, n, entry, pops):
    """The rows a walk scores, in order, restated from its pop sequence and the element image (hnswalg.cpp:55-97): the entry
    point, then for every popped element its links in order that were not visited before.  What the kernels' evaluation trace
    (hnsw_gpu_search_traced_dev) must equal."""
    esz = int(meta.size_data_per_element)
    img = np.frombuffer(np.ascontiguousarray(raw)[: n * esz], np.uint8).reshape(n, esz)
    links = img[:, : (int(meta.maxM) + 1) * 4].copy().view(np.uint32)        # [count | links]
    seen = {int(entry)}
    out = [int(entry)]
    for p in pops:
        cnt = int(links[p, 0])
        for t in links[p, 1:1 + cnt]:
            t = int(t)
            if t not in seen:
                seen.add(t)
                out.append(t)
    return np.asarray(out, np.uint32)


def foreign_toolchain(what: str):
    """The reference-order debug arithmetic (HNSW_GPU_REF_ORDER=1, csrc/device_dist.h score_rows_ref) restates the summation order of
    ONE build of distfunc.c: gcc 11.4 -Ofast, the toolchain of this image, read off its disassembly.  Where oracle/_ref was built by
    a compiler that vectorises differently the direct device-vs-reference comparison cannot hold — and must not disappear silently:
    the test FAILS unless PGEMB_FOREIGN_REF_TOOLCHAIN=1 says that this is known (then it is skipped, loudly, and what remains is
    the classification chain of test_gpu_fullsize.py: every differing id list explained by a decision within 1e-5)."""
    import os
    import pytest
    msg = (f"{what}: this host's oracle/_ref build sums in another order than the one score_rows_ref restates (a different compiler?).  "
           "Set PGEMB_FOREIGN_REF_TOOLCHAIN=1 to acknowledge it and skip the direct comparison")
    if os.environ.get("PGEMB_FOREIGN_REF_TOOLCHAIN") == "1":
        pytest.skip(msg)
    pytest.fail(msg)
