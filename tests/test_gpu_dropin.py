"""The drop-in boundary: the reference's host side (storage callbacks, oracle/flat_host.c) linked
against libembedding_gpu.so instead of hnswalg.o + distfunc.o.  Everything goes through the four
symbols of embedding.h:46-47,55-56."""
import ctypes as C

import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from pg_embedding_amd import build as B
from pg_embedding_amd._lib import shim_lib
from pg_embedding_amd.datasets import gmm
from test_gpu_build import live_image

pytestmark = pytest.mark.gpu


def host(dim, m, efc, efs, func):
    return oracle.FlatHostIndex(B.SHIM_LIB, dim, m, efc, efs, func)


def test_hnsw_dist_func_symbol():
    L = shim_lib()
    a = np.array([1, 2, 3], np.float32)
    b = np.array([3, 3, 3], np.float32)
    f = lambda func: L.hnsw_dist_func(func, a.ctypes.data_as(C.POINTER(C.c_float)),
                                      b.ctypes.data_as(C.POINTER(C.c_float)), 3)
    L.hnsw_init_dist_func()
    assert abs(f(0) - 2.236068) < 1e-6 and abs(f(1) - 0.0741799) < 1e-6 and f(2) == 3.0


@pytest.mark.parametrize("func", [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN])
def test_hnsw_search_symbol_mirrors_through_callbacks(func):
    dim, m, n, efs = 48, 6, 1500, 40
    X = gmm(n, dim, k=30, seed=40 + func)
    Q = gmm(30, dim, k=30, seed=40 + func, stream=1)
    port = oracle.PortIndex(dim, m, 32, efs, func)
    port.add(X, np.arange(n, dtype=np.uint64) + 10)
    for i in (3, 77, 500):
        port.set_deleted(i)
    h = host(dim, m, 32, efs, func)
    h.load_raw(port.raw(), n)
    for q in range(30):
        got = h.search(Q[q], efs)                 # hnsw_search -> snapshot via hnsw_begin_read -> kernel
        want = port.search(Q[q], efs)[0]
        assert (got == want).all()
    # efSearch doubling of the scan (embedding.c:334): the caller mutates meta between calls
    got = h.search(Q[0], 2 * efs)
    assert (got == port.search(Q[0], 2 * efs)[0]).all()


def test_attached_mirror_is_used_and_empty_index_works():
    L = shim_lib()
    dim, m, n, efs = 32, 5, 800, 32
    X = gmm(n, dim, k=20, seed=77)
    port = oracle.PortIndex(dim, m, 24, efs, pg.DIST_L2)
    port.add(X)
    h = host(dim, m, 24, efs, pg.DIST_L2)
    assert h.search(X[0], efs).size == 0          # gh-2: empty index -> true, 0 rows
    h.load_raw(port.raw(), n)
    mirror = C.c_void_p()
    assert L.hnsw_gpu_shim_snapshot(h.meta, C.byref(mirror)) == 0
    assert L.hnsw_gpu_shim_attach(h.meta, mirror) == 0
    for q in range(20):
        assert (h.search(X[q * 7], efs) == port.search(X[q * 7], efs)[0]).all()
    # the attached mirror is really what answers: flag an element on the device only
    first = int(h.search(X[5], efs)[0])
    from pg_embedding_amd._lib import gpu_lib
    assert gpu_lib().hnsw_gpu_index_set_deleted(mirror, first, 1) == 0
    assert first not in h.search(X[5], efs).tolist()
    assert L.hnsw_gpu_shim_detach(h.meta) == 0
    assert first in h.search(X[5], efs).tolist()  # back to mirroring the host, where it is alive
    gpu_lib().hnsw_gpu_index_destroy(mirror)


@pytest.mark.parametrize("attached", [False, True])
def test_hnsw_bind_point_symbol_builds_the_reference_graph(attached):
    """Insert by insert through the shim (device insert in serial mode + write-back through
    hnsw_begin_write): the host's pages end up with the oracle's link lists."""
    L = shim_lib()
    dim, m, efc, n = 24, 4, 16, 260 if not attached else 700
    X = gmm(n, dim, k=12, seed=91)
    labels = np.arange(n, dtype=np.uint64) + 1000
    port = oracle.PortIndex(dim, m, efc, 32, pg.DIST_L2)
    port.add(X, labels)
    h = host(dim, m, efc, 32, pg.DIST_L2)
    mirror = C.c_void_p()
    if attached:
        h.add(X[:1], labels[:1])
        assert L.hnsw_gpu_shim_snapshot(h.meta, C.byref(mirror)) == 0
        assert L.hnsw_gpu_shim_attach(h.meta, mirror) == 0
        h.add(X[1:], labels[1:])                  # flat_add = append + hnsw_bind_point, embedding.c:606-701
    else:
        h.add(X, labels)
    meta = pg.make_meta(dim, m, efc, 32, pg.DIST_L2)
    assert (live_image(h.raw(), meta, n) == live_image(port.raw(), meta, n)).all()
    got = h.search(X[3], 32)
    assert (got == port.search(X[3], 32)[0]).all()
    if attached:
        L.hnsw_gpu_shim_detach(h.meta)
        from pg_embedding_amd._lib import gpu_lib
        gpu_lib().hnsw_gpu_index_destroy(mirror)


def test_page_tail_holes_in_element_numbers():
    """dims=3, m=3: elems_per_page computes 157 but 156 fit -> idx 156, 313, ...
    never exist.  Build with the reference over such a host, search through the shim."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    dim, m, n, efs = 3, 3, 700, 20
    rng = np.random.default_rng(5)
    X = rng.integers(0, 9, (n, dim)).astype(np.float32)
    ref = oracle.RefIndex(dim, m, 16, efs, pg.DIST_L2)
    ref.set_page_real(156)
    ref.add(X)
    assert ref.idx_end == n + (n - 1) // 156      # holes shifted the numbering
    h = host(dim, m, 16, efs, pg.DIST_L2)
    h.set_page_real(156)
    h.load_raw(ref.raw(), n)
    for q in range(25):
        v = rng.integers(0, 9, dim).astype(np.float32)
        assert (h.search(v, efs) == ref.search(v, efs)).all()
    # and inserts through the shim continue across a page boundary
    more = rng.integers(0, 9, (40, dim)).astype(np.float32)
    ref.add(more)
    h.add(more)
    for q in range(10):
        v = rng.integers(0, 9, dim).astype(np.float32)
        assert (h.search(v, efs) == ref.search(v, efs)).all()


def test_index_scan_follows_hnsw_gettuple():
    """embedding.c:284-370 restated over the reference binary (oracle/_ref hnsw_search) vs
    pg_embedding_amd.scan.IndexScan over the device search: same TIDs in the same order,
    including the efSearch-doubling re-scan and its de-duplication."""
    from pg_embedding_amd.scan import IndexScan
    dim, m, n, efs = 16, 4, 900, 8
    X = sift_like_rows(n, dim)
    port = oracle.PortIndex(dim, m, 16, efs, pg.DIST_L2)
    port.add(X)
    ix = pg.GpuIndex.from_flat(pg.make_meta(dim, m, 16, efs, pg.DIST_L2), port.raw(), n)

    def reference_scan(q, limit):
        ef = efs
        res = list(port.search(q, ef)[0].tolist())
        no_more = len(res) < ef
        out, curr = [], 0
        while len(out) < limit:
            if curr >= len(res):
                if no_more:
                    break
                ef *= 2
                r = port.search(q, ef)[0].tolist()
                if len(r) <= len(res):
                    break
                no_more = len(r) < ef
                seen = set(res)
                res.extend(x for x in r if x not in seen)
                if curr >= len(res):
                    break
            out.append(res[curr])
            curr += 1
        return out

    for qi in range(10):
        q = X[qi * 13] + 0.25
        want = reference_scan(q, 100)
        got = []
        for lab in IndexScan(ix, q, efs):
            got.append(lab)
            if len(got) == 100:
                break
        assert got == want and len(got) == 100
    # LIMIT larger than the table: the scan doubles efSearch past the index size (8 -> ... -> 1024 > 900)
    # and ends when a search comes back short; every reachable row is returned exactly once
    q = X[7] + 0.25
    want = reference_scan(q, 10 ** 9)
    got = list(IndexScan(ix, q, efs))
    assert got == want and len(set(got)) == len(got) and len(got) > 800
    ix.close()


def sift_like_rows(n, dim):
    from pg_embedding_amd.datasets import sift_like
    return sift_like(n, dim, k=10, seed=6)


def test_c_host_linked_against_the_gpu_library(tmp_path):
    """A C host (storage callbacks + calls to the four symbols only), linked once against the
    reference objects (oracle/_ref) and once against libembedding_gpu.so: same output."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = [os.path.join(root, "tests", "dropin_c", "dropin_demo.c"), os.path.join(root, "oracle", "flat_host.c")]
    inc = os.path.join(root, "include")
    lib = os.path.join(root, "pg_embedding_amd", "lib")
    gpu_exe = str(tmp_path / "demo_gpu")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-I", inc] + src + ["-o", gpu_exe, "-L", lib, "-lembedding_gpu",
                    "-lhnsw_gpu", f"-Wl,-rpath,{lib}", "-lpthread", "-lm"], check=True)
    args = ["400", "24", "4", "16", "12", "15"]
    got = subprocess.run([gpu_exe] + args, capture_output=True, text=True, check=True).stdout
    assert got.count("\n") == 15 and "d0=0.000000" in got
    refdir = os.path.join(root, "oracle", "_ref")
    if os.path.exists(os.path.join(refdir, "hnswalg.o")):
        ref_exe = str(tmp_path / "demo_ref")
        objs = []
        for i, c in enumerate(src):
            o = str(tmp_path / f"host{i}.o")
            subprocess.run(["gcc", "-O2", "-std=gnu11", "-I", inc, "-c", c, "-o", o], check=True)
            objs.append(o)
        subprocess.run(["g++"] + objs + [os.path.join(refdir, "hnswalg.o"), os.path.join(refdir, "distfunc.o"),
                        "-o", ref_exe, "-lpthread", "-lm"], check=True)
        want = subprocess.run([ref_exe] + args, capture_output=True, text=True, check=True).stdout
        assert got == want
