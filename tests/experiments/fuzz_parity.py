"""Randomised parity sweep: device search / serial insert vs the CPU oracle over random shapes
(dims that are not multiples of 4/16/64, odd m, link lists longer than a wave, vacuumed labels,
every metric, ef below/above the register-form limits).  Prints one line per case; exits non-zero on
the first mismatch.   usage: fuzz_parity.py <cases> [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pg_embedding_amd import watchdog; watchdog.arm()      # --timeout SECONDS (default 900): a hung device run costs one case, not the round
import numpy as np
import oracle
import pg_embedding_amd as pg
from pg_embedding_amd.datasets import gmm


def one_case(rng, idx):
    func = int(rng.integers(0, 3))
    dim = int(rng.choice([1, 2, 3, 5, 7, 16, 31, 33, 64, 65, 100, 127, 128, 129, 200, 256, 300, 384, 500, 513, 768, 769, 1000, 1536, 1999]))
    m = int(rng.choice([1, 2, 3, 4, 5, 8, 12, 16, 24, 32, 33, 50, 100]))
    while (2 * m + 1) * 4 + dim * 4 + 8 + 4 > 8192 - 28:      # must fit a Postgres page, embedding.c:229-231
        m = max(1, m // 2)
    n = int(rng.integers(2, 4000 if dim <= 512 else 1500))
    efc = int(rng.choice([1, 4, 16, 40, 100]))
    ef = int(rng.choice([1, 2, 7, 16, 64, 100, 128, 129, 200, 256, 257, 400, 512, 513, 700, 5000]))
    k = int(rng.integers(1, 40))
    X = gmm(n, dim, k=k, sigma=float(rng.choice([0.05, 0.3, 1.0])), seed=1000 + idx)
    if rng.random() < 0.3:
        X = np.rint(X * 4).astype(np.float32)          # many exact ties
    if rng.random() < 0.3 and n > 10:
        X[n // 2:n // 2 + n // 10] = X[:n // 10]        # duplicate rows
    if func == 1:
        # zero vectors give NaN cosine distances (0/0, distfunc.c:144): outside the parity contract
        # (SURVEY.md §7 "Cosine numerics"); NaN robustness has its own test
        X[(X * X).sum(axis=1) == 0] = 1.0
    labels = rng.permutation(n).astype(np.uint64) + np.uint64(rng.integers(0, 1 << 40))
    port = oracle.PortIndex(dim, m, efc, ef, func)
    port.add(X, labels)
    ndel = int(rng.integers(0, max(1, n // 3)))
    for i in rng.choice(n, ndel, replace=False):
        port.set_deleted(int(i))
    meta = pg.make_meta(dim, m, efc, ef, func)
    ix = pg.GpuIndex.from_flat(meta, port.raw(), n)
    nq = int(rng.choice([1, 3, 16, 17, 60, 300, 700]))       # <= 16: polled zero-copy call; > one per CU: the 5-waves narrow-row kernel
    Q = gmm(nq, dim, k=k, seed=1000 + idx, stream=1)
    if func == 1:
        Q[(Q * Q).sum(axis=1) == 0] = 1.0
    L, D, Cn = ix.search(Q, ef)
    W = port.search_many(Q, ef)
    ok = (Cn == W["counts"]).all()
    for q in range(nq):
        c = int(W["counts"][q])
        ok = ok and (L[q, :c] == W["labels"][q, :c]).all() and \
            (D[q, :c].view(np.uint32) == W["dists"][q, :c].view(np.uint32)).all()
    # the walk itself: pop sequence and evaluation count of one query, one-step and three-step form (round 2)
    tl, td, tp, tev = ix.search_trace(Q[0], ef)
    wl, wd, wp, wev = port.search_trace(Q[0], min(ef, n))
    _, _, tp2, _, _ = ix.search_trace_polled(Q[0], ef)
    ok = ok and len(tp) == len(wp) and (tp == wp).all() and tev == wev and (tl == wl).all() and len(tp2) == len(wp) and (tp2 == wp).all()
    ix.close()
    # serial device insert == oracle graph (small prefix to keep it quick)
    nb = min(n, 250)
    ix2 = pg.GpuIndex.empty(meta, nb)
    ix2.append(X[:nb], labels[:nb])
    ix2.link(0, nb, max_batch=1)
    p2 = oracle.PortIndex(dim, m, efc, ef, func)
    p2.add(X[:nb], labels[:nb])
    a = ix2.export_flat().reshape(nb, -1)
    b = p2.raw().reshape(nb, -1).copy()
    lw = b[:, :meta.offset_data].copy().view(np.uint32)
    for e in range(nb):
        lw[e, 1 + lw[e, 0]:] = 0
    b[:, :meta.offset_data] = lw.view(np.uint8)
    okb = bool((a == b).all())
    ix2.close()
    print(f"case {idx}: func={func} dim={dim} m={m} n={n} efc={efc} ef={ef} nq={nq} del={ndel} search={'ok' if ok else 'MISMATCH'} build={'ok' if okb else 'MISMATCH'}", flush=True)
    return bool(ok) and okb


if __name__ == "__main__":
    cases = int(sys.argv[1])
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for i in range(cases):
        if not one_case(rng, i + 100000 * seed):
            bad += 1
            break
    print(f"{cases} cases, seed {seed}: {'ALL OK' if bad == 0 else 'FAILED'}")
    sys.exit(1 if bad else 0)
