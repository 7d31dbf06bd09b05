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
