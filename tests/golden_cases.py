"""Loader for the committed fixtures (tests/golden/)."""
import importlib.util
import json
import os

import numpy as np

import oracle

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FUNC = {"l2": 0, "cosine": 1, "manhattan": 2}


def knn_expected():
    with open(os.path.join(HERE, "knn_expected.json")) as f:
        return json.load(f)


def _mk():
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(HERE, "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ref_cases():
    """yield (name, func, n, dim, m, efc, ef, X, Q, fixture-dict) for every committed case"""
    mk = _mk()
    z = np.load(os.path.join(HERE, "ref_cases.npz"))
    for name, func, n, dim, m, efc, ef, kind in mk.CASES:
        X, Q = mk.case_data(name, n, dim, kind)
        fx = {k[len(name) + 1:]: z[k] for k in z.files if k.startswith(name + "_")}
        yield name, func, n, dim, m, efc, ef, X, Q, fx


def image_from_links(links, X, labels=None):
    """element images [count|links|vector|label] from a link table and rows"""
    n, dim = X.shape
    m = (links.shape[1] - 1) // 2
    esz = oracle.elem_size(dim, m)
    raw = np.zeros((n, esz), np.uint8)
    raw[:, :links.shape[1] * 4] = links.view(np.uint8).reshape(n, -1)
    raw[:, links.shape[1] * 4:links.shape[1] * 4 + dim * 4] = X.view(np.uint8).reshape(n, -1)
    lab = np.arange(n, dtype=np.uint64) if labels is None else labels
    raw[:, -8:] = lab.view(np.uint8).reshape(n, 8)
    return raw.ravel()
