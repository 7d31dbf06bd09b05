"""Generate tests/golden/ref_cases.npz from the UNMODIFIED reference (oracle/_ref).

Run in the build container, where /root/reference exists:
    python tests/golden/make_golden.py
For each case the reference builds a graph over seeded data (hnsw_bind_point) and answers
seeded queries (hnsw_search); the fixture keeps the link lists, the returned labels and a
sample of reference distances so that the C restatement and the device path can be checked
against the real reference where /root/reference is absent (the GPU box).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from pg_embedding_amd.datasets import gmm, sift_like  # noqa: E402

CASES = [
    # name, func, n, dim, m, efc, ef, data
    ("l2_128", 0, 2000, 128, 8, 64, 64, "gmm"),
    ("cos_96", 1, 1500, 96, 6, 48, 64, "gmm"),
    ("man_64", 2, 1500, 64, 6, 48, 64, "gmm"),
    ("l2_768", 0, 800, 768, 16, 64, 128, "gmm"),
    ("l2_sift", 0, 2000, 128, 8, 64, 64, "sift"),
]


def case_data(name, n, dim, kind, nq=40):
    seed = sum(map(ord, name))
    if kind == "sift":
        return sift_like(n, dim, k=40, seed=seed), sift_like(nq, dim, k=40, seed=seed, stream=1)
    return gmm(n, dim, k=40, seed=seed), gmm(nq, dim, k=40, seed=seed, stream=1)


def links_of(raw, n, dim, m):
    esz = oracle.elem_size(dim, m)
    return raw.reshape(n, esz)[:, :(2 * m + 1) * 4].copy().view(np.uint32)


def main():
    assert oracle.have_ref(), "needs oracle/_ref (make -C oracle ref)"
    out = {}
    for name, func, n, dim, m, efc, ef, kind in CASES:
        X, Q = case_data(name, n, dim, kind)
        ref = oracle.RefIndex(dim, m, efc, ef, func)
        ref.add(X)
        r = ref.search_many(Q, ef)
        out[name + "_links"] = links_of(ref.raw(), n, dim, m)
        out[name + "_labels"] = r["labels"]
        out[name + "_counts"] = r["counts"]
        out[name + "_evals"] = r["evals"]
        out[name + "_hops"] = r["hops"]
        out[name + "_dist0"] = oracle.ref_dist_many(func, Q[0], X)       # reference distances, query 0
        print(name, "evals/query", r["evals"].mean(), "hops/query", r["hops"].mean())
    np.savez_compressed(os.path.join(HERE, "ref_cases.npz"), **out)


if __name__ == "__main__":
    main()
