"""Seeded synthetic inputs of the shapes BASELINE.json names.

The recall-gated metric needs clustered data: i.i.d. Gaussian vectors give recall@10 of
only 0.65-0.75 at efsearch=128 with this single-layer graph, so the base
generator is a Gaussian mixture: K centres ~ N(0, I), point = centre + sigma * N(0, I);
queries come from the same mixture with a different stream.
"""
from __future__ import annotations

import numpy as np


def gmm(n: int, dim: int, k: int = 1000, sigma: float = 0.3, seed: int = 42, stream: int = 0) -> np.ndarray:
    """numpy version (small cases, parity tests)."""
    crng = np.random.default_rng([seed, 0xC0])
    centres = crng.standard_normal((k, dim), dtype=np.float32)
    rng = np.random.default_rng([seed, 1 + stream])
    which = rng.integers(0, k, n)
    x = centres[which] + np.float32(sigma) * rng.standard_normal((n, dim), dtype=np.float32)
    return np.ascontiguousarray(x, dtype=np.float32)


def gmm_chunks(n: int, dim: int, k: int = 1000, sigma: float = 0.3, seed: int = 42, stream: int = 0, chunk: int = 1 << 16):
    """gmm() in pieces: yields (first_row, rows) with exactly the bytes gmm(n, ...) returns for those rows (the generator's normal stream
    is sequential, so consecutive draws concatenate to the one big draw) without ever holding the table and its temporaries on the host:
    1M x 768 is 3 GB, and gmm() peaks at three times that."""
    crng = np.random.default_rng([seed, 0xC0])
    centres = crng.standard_normal((k, dim), dtype=np.float32)
    rng = np.random.default_rng([seed, 1 + stream])
    which = rng.integers(0, k, n)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        x = centres[which[a:b]] + np.float32(sigma) * rng.standard_normal((b - a, dim), dtype=np.float32)
        yield a, np.ascontiguousarray(x, dtype=np.float32)


def sift_like(n: int, dim: int = 128, k: int = 256, seed: int = 42, stream: int = 0) -> np.ndarray:
    """SIFT-1M stand-in (the real files are not available offline): clustered non-negative
    integers in [0, 218] stored as fp32, which makes every L2^2 an exact integer."""
    x = gmm(n, dim, k, 0.35, seed, stream)
    x = np.clip(np.rint(40.0 + 35.0 * x), 0, 218)
    return np.ascontiguousarray(x, dtype=np.float32)


def gmm_torch(n: int, dim: int, k: int = 1000, sigma: float = 0.3, seed: int = 42, stream: int = 0,
              device="cuda", chunk: int = 1 << 18, distinct_clusters: bool = False):
    """Device generator for the full-size configs (1M x 768 = 3 GB is produced in HBM).
    distinct_clusters: point i comes from a cluster of its own (a random permutation of the k clusters; needs n <= k) —
    queries that share no home cluster."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed * 1000003 + 0xC0)
    centres = torch.randn((k, dim), generator=g, device=device, dtype=torch.float32)
    g.manual_seed(seed * 1000003 + 1 + stream)
    out = torch.empty((n, dim), device=device, dtype=torch.float32)
    perm = None
    if distinct_clusters:
        assert n <= k, "distinct_clusters needs at least as many clusters as points"
        perm = torch.randperm(k, generator=g, device=device)
    for i in range(0, n, chunk):
        m = min(chunk, n - i)
        which = perm[i:i + m] if perm is not None else torch.randint(0, k, (m,), generator=g, device=device)
        out[i:i + m] = centres[which] + sigma * torch.randn((m, dim), generator=g, device=device,
                                                            dtype=torch.float32)
    return out


def recall_at_k(found: np.ndarray, truth: np.ndarray, k: int = 10) -> float:
    """|first k returned ∩ exact top-k| / k averaged over queries."""
    found = np.asarray(found)[:, :k]
    truth = np.asarray(truth)[:, :k]
    hit = 0
    for f, t in zip(found, truth):
        hit += len(set(f.tolist()) & set(t.tolist()))
    return hit / float(k * len(truth))
