"""ctypes bindings for the oracle libraries — TEST INFRASTRUCTURE ONLY (see __init__)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

DIST_L2, DIST_COSINE, DIST_MANHATTAN = 0, 1, 2   # embedding.h:22-26

_HERE = os.path.dirname(os.path.abspath(__file__))
_PORT_SO = os.path.join(_HERE, "_build", "libhnsw_port.so")
_HOST_SO = os.path.join(_HERE, "_build", "libflat_host.so")
_REF_SO = os.path.join(_HERE, "_ref", "libpgemb_ref.so")

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


def build_oracle(quiet: bool = True) -> None:
    """Compile the C restatement (+ flat host) and, when /root/reference exists,
    the reference-derived oracle/_ref.  Building the checker is not using it."""
    cmd = ["make", "-C", _HERE, "all"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if not quiet:
        print(r.stdout)


def have_ref() -> bool:
    return os.path.exists(_REF_SO)


def elem_size(dim: int, m: int) -> int:
    """Bytes per element image (embedding.c:225-228)."""
    return (2 * m + 1) * 4 + dim * 4 + 8


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, typ):
    return a.ctypes.data_as(typ)


# --------------------------------------------------------------------------- port
_port_lib = None


def _port():
    global _port_lib
    if _port_lib is None:
        if not os.path.exists(_PORT_SO):
            build_oracle()
        L = C.CDLL(_PORT_SO)
        L.port_dist.restype = C.c_float
        L.port_dist.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
        L.port_dist_many.restype = None
        L.port_dist_many.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t, C.c_size_t, _f32p]
        L.port_create.restype = C.c_void_p
        L.port_create.argtypes = [C.c_size_t] * 4 + [C.c_int, C.c_size_t]
        L.port_destroy.argtypes = [C.c_void_p]
        L.port_count.restype = C.c_size_t
        L.port_count.argtypes = [C.c_void_p]
        L.port_data.restype = C.c_void_p
        L.port_data.argtypes = [C.c_void_p]
        L.port_elem_size.restype = C.c_size_t
        L.port_elem_size.argtypes = [C.c_void_p]
        L.port_load_raw.restype = C.c_int
        L.port_load_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.port_set_deleted.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.port_add_many.restype = C.c_long
        L.port_add_many.argtypes = [C.c_void_p, _f32p, _u64p, C.c_size_t]
        L.port_search.restype = C.c_int
        L.port_search.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u64p, _f32p,
                                  C.POINTER(C.c_size_t), _u32p, _u32p]
        L.port_search_base.restype = C.c_int
        L.port_search_base.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u32p, _f32p,
                                       C.POINTER(C.c_size_t), _u32p, _u32p]
        L.port_search_trace.restype = C.c_int
        L.port_search_trace.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_int, _u64p, _f32p, C.POINTER(C.c_size_t),
                                        _u32p, _u32p, C.c_size_t, _u32p]
        L.port_search_many.restype = C.c_double
        L.port_search_many.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, C.c_int,
                                       _u64p, _f32p, _u32p, _u32p, _u32p]
        L.port_search_many_m.restype = C.c_double
        L.port_search_many_m.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, C.c_int,
                                         _u64p, _f32p, _u32p, _u32p, _u32p, _f32p,
                                         C.POINTER(C.c_int32), _f32p]
        L.port_set_dist_fn.restype = None
        L.port_set_dist_fn.argtypes = [C.c_void_p, C.c_void_p]
        L.port_set_dist2_fn.restype = None
        L.port_set_dist2_fn.argtypes = [C.c_void_p, C.c_void_p]
        L.port_add_shadowed.restype = C.c_long
        L.port_add_shadowed.argtypes = [C.c_void_p, _f32p, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        _port_lib = L
    return _port_lib


def port_dist(func: int, a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(_port().port_dist(func, _ptr(a, _f32p), _ptr(b, _f32p), a.size))


def port_dist_many(func: int, q, rows) -> np.ndarray:
    q, rows = _f32(q), _f32(rows)
    out = np.empty(rows.shape[0], np.float32)
    _port().port_dist_many(func, _ptr(q, _f32p), _ptr(rows, _f32p), rows.shape[0], q.size,
                           _ptr(out, _f32p))
    return out


class PortIndex:
    """The C restatement over its own element image (same bytes as the host's)."""

    def __init__(self, dim: int, m: int, efc: int = 16, efs: int = 64, func: int = DIST_L2,
                 capacity: int = 0):
        self.L = _port()
        self.dim, self.m, self.efc, self.efs, self.func = dim, m, efc, efs, func
        self.h = self.L.port_create(dim, m, efc, efs, func, capacity)
        if not self.h:
            raise MemoryError("port_create")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.port_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def count(self) -> int:
        return int(self.L.port_count(self.h))

    @property
    def elem_size(self) -> int:
        return int(self.L.port_elem_size(self.h))

    def add(self, vecs, labels=None) -> None:
        vecs = _f32(vecs).reshape(-1, self.dim)
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint64)
            lp = _ptr(labels, _u64p)
        r = self.L.port_add_many(self.h, _ptr(vecs, _f32p), lp, vecs.shape[0])
        if r < 0:
            raise RuntimeError(f"port_add_many failed ({r})")

    def raw(self) -> np.ndarray:
        n = self.count * self.elem_size
        buf = (C.c_uint8 * n).from_address(self.L.port_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8).copy()

    def raw_view(self) -> np.ndarray:
        """The element image itself (writable; valid until the next add)."""
        n = self.count * self.elem_size
        buf = (C.c_uint8 * n).from_address(self.L.port_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8)

    def add_shadowed(self, vec, label: int):
        """One serial insert (hnsw_bind_point, hnswalg.cpp:117-232, 279-291) with its decision record under the shadow
        arithmetic (shadow_reference_distances): (idx, div_kind, div_margin) — div_kind 0: the reference's arithmetic
        builds the same lists from the same state."""
        vec = _f32(vec)
        kind, margin = C.c_int32(0), C.c_float(0)
        r = self.L.port_add_shadowed(self.h, _ptr(vec, _f32p), int(label), C.byref(kind), C.byref(margin))
        if r < 0:
            raise RuntimeError(f"port_add_shadowed failed ({r})")
        return int(r), int(kind.value), float(margin.value)

    def load_raw(self, raw: np.ndarray, n: int) -> None:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.size == n * self.elem_size
        if self.L.port_load_raw(self.h, raw.ctypes.data, n) != 0:
            raise MemoryError("port_load_raw")

    def set_deleted(self, idx: int, deleted: bool = True) -> None:
        self.L.port_set_deleted(self.h, idx, int(deleted))

    def use_reference_distances(self, on: bool = True) -> None:
        """Score with the reference's own hnsw_dist_func (oracle/_ref) instead of the canonical-order
        restatement: the traversal restated in hnsw_port.c must then reproduce the reference's results
        exactly, which isolates the summation order as the only difference between device and reference."""
        fn = C.cast(_ref().hnsw_dist_func, C.c_void_p) if on else None
        self.L.port_set_dist_fn(self.h, fn)

    def shadow_reference_distances(self, on: bool = True) -> None:
        """Walk in the canonical arithmetic, but also score every evaluation with the reference's
        hnsw_dist_func and record, per query, the first decision the reference's values would have taken
        differently (search_many: 'div_kind', 'div_margin').  div_kind == 0 proves the reference returns the
        same ids; a mismatching query has div_kind != 0 and div_margin says how close that call was."""
        fn = C.cast(_ref().hnsw_dist_func, C.c_void_p) if on else None
        self.L.port_set_dist2_fn(self.h, fn)

    def search(self, q, ef: Optional[int] = None):
        """hnsw_search semantics: (labels, dists, evals, hops), ascending by (dist, label)."""
        ef = ef or self.efs
        q = _f32(q)
        lab = np.empty(ef, np.uint64)
        dst = np.empty(ef, np.float32)
        n = C.c_size_t(0)
        ev, hp = C.c_uint32(0), C.c_uint32(0)
        self.L.port_search(self.h, _ptr(q, _f32p), ef, _ptr(lab, _u64p), _ptr(dst, _f32p),
                           C.byref(n), C.byref(ev), C.byref(hp))
        return lab[:n.value].copy(), dst[:n.value].copy(), ev.value, hp.value

    def search_base(self, q, ef: int):
        """searchBaseLayer result: (idx, dists) ascending by (dist, idx)."""
        q = _f32(q)
        idx = np.empty(ef, np.uint32)
        dst = np.empty(ef, np.float32)
        n = C.c_size_t(0)
        ev, hp = C.c_uint32(0), C.c_uint32(0)
        self.L.port_search_base(self.h, _ptr(q, _f32p), ef, _ptr(idx, _u32p), _ptr(dst, _f32p),
                                C.byref(n), C.byref(ev), C.byref(hp))
        return idx[:n.value].copy(), dst[:n.value].copy(), ev.value, hp.value

    def search_trace(self, q, ef: int, base: bool = False, pops_cap: int = 1 << 16):
        """hnsw_search (or searchBaseLayer) with the walk's pop sequence: (labels or idx, dists, pops, evals)."""
        q = _f32(q)
        lab = np.empty(ef, np.uint64)
        dst = np.empty(ef, np.float32)
        pops = np.empty(pops_cap, np.uint32)
        n = C.c_size_t(0)
        ev, npops = C.c_uint32(0), C.c_uint32(0)
        self.L.port_search_trace(self.h, _ptr(q, _f32p), ef, int(base), _ptr(lab, _u64p), _ptr(dst, _f32p), C.byref(n),
                                 C.byref(ev), _ptr(pops, _u32p), pops_cap, C.byref(npops))
        return lab[:n.value].copy(), dst[:n.value].copy(), pops[:min(npops.value, pops_cap)].copy(), ev.value

    def search_many(self, Q, ef: Optional[int] = None, nthreads: int = 1) -> dict:
        ef = ef or self.efs
        Q = _f32(Q).reshape(-1, self.dim)
        nq = Q.shape[0]
        lab = np.zeros((nq, ef), np.uint64)
        dst = np.full((nq, ef), np.inf, np.float32)
        cnt = np.zeros(nq, np.uint32)
        ev = np.zeros(nq, np.uint32)
        hp = np.zeros(nq, np.uint32)
        mg = np.zeros(nq, np.float32)
        dk = np.zeros(nq, np.int32)
        dm = np.full(nq, np.inf, np.float32)
        sec = self.L.port_search_many_m(self.h, _ptr(Q, _f32p), nq, ef, nthreads, _ptr(lab, _u64p),
                                        _ptr(dst, _f32p), _ptr(cnt, _u32p), _ptr(ev, _u32p),
                                        _ptr(hp, _u32p), _ptr(mg, _f32p), _ptr(dk, C.POINTER(C.c_int32)),
                                        _ptr(dm, _f32p))
        # margins[q]: smallest relative gap between two different elements' distances over every
        # comparison that steered query q's walk or its output order (hnsw_port.c, PortStats)
        # div_kind/div_margin: only meaningful after shadow_reference_distances() (0 / inf otherwise)
        return dict(labels=lab, dists=dst, counts=cnt, evals=ev, hops=hp, margins=mg, div_kind=dk,
                    div_margin=dm, seconds=sec)


# ------------------------------------------------------------------- flat host family
class HnswMetadata(C.Structure):
    """ctypes image of embedding.h:28-42."""
    _fields_ = [(n, C.c_size_t) for n in (
        "dim", "data_size", "offset_data", "offset_label", "size_data_per_element",
        "elems_per_page", "M", "maxM", "efConstruction", "efSearch")] + [
        ("enterpoint_node", C.c_uint32), ("dist_func", C.c_int)]


def _bind_flat(L):
    L.flat_create.restype = C.c_void_p
    L.flat_create.argtypes = [C.c_size_t] * 4 + [C.c_int, C.c_size_t]
    L.flat_destroy.argtypes = [C.c_void_p]
    L.flat_meta.restype = C.POINTER(HnswMetadata)
    L.flat_meta.argtypes = [C.c_void_p]
    L.flat_count.restype = C.c_size_t
    L.flat_count.argtypes = [C.c_void_p]
    L.flat_data.restype = C.c_void_p
    L.flat_data.argtypes = [C.c_void_p]
    L.flat_elem_size.restype = C.c_size_t
    L.flat_elem_size.argtypes = [C.c_void_p]
    L.flat_set_ef_search.argtypes = [C.c_void_p, C.c_size_t]
    L.flat_set_page_real.restype = C.c_int
    L.flat_set_page_real.argtypes = [C.c_void_p, C.c_size_t]
    L.flat_idx_end.restype = C.c_size_t
    L.flat_idx_end.argtypes = [C.c_void_p]
    L.flat_set_ef_construction.argtypes = [C.c_void_p, C.c_size_t]
    L.flat_append.restype = C.c_long
    L.flat_append.argtypes = [C.c_void_p, _f32p, C.c_uint64]
    L.flat_add.restype = C.c_long
    L.flat_add.argtypes = [C.c_void_p, _f32p, C.c_uint64]
    L.flat_add_many.restype = C.c_long
    L.flat_add_many.argtypes = [C.c_void_p, _f32p, _u64p, C.c_size_t]
    L.flat_load_raw.restype = C.c_int
    L.flat_load_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.flat_set_deleted.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.flat_dist.restype = C.c_float
    L.flat_dist.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
    L.flat_dist_many.restype = None
    L.flat_dist_many.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t, C.c_size_t, _f32p]
    L.flat_counters_reset.restype = None
    L.flat_counters_get.argtypes = [C.POINTER(C.c_uint64 * 4)]
    L.flat_search.restype = C.c_int
    L.flat_search.argtypes = [C.c_void_p, _f32p, C.c_size_t, _u64p, C.POINTER(C.c_size_t)]
    L.flat_search_many.restype = C.c_double
    L.flat_search_many.argtypes = [C.c_void_p, _f32p, C.c_size_t, C.c_size_t, C.c_int,
                                   _u64p, _u32p, _u32p, _u32p]
    return L


class _FlatIndexBase:
    """Python face of oracle/flat_host.c, whichever hot path is linked behind it."""
    L = None

    def __init__(self, dim: int, m: int, efc: int = 16, efs: int = 64, func: int = DIST_L2,
                 capacity: int = 0):
        self.dim, self.m, self.efc, self.efs, self.func = dim, m, efc, efs, func
        self.h = self.L.flat_create(dim, m, efc, efs, func, capacity)
        if not self.h:
            raise MemoryError("flat_create")

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.flat_destroy(self.h)
                self.h = None
        except Exception:
            pass

    @property
    def count(self) -> int:
        return int(self.L.flat_count(self.h))

    @property
    def elem_size(self) -> int:
        return int(self.L.flat_elem_size(self.h))

    def set_page_real(self, page_real: int) -> None:
        """Give element numbers tail-of-page holes; only on an empty index."""
        if self.L.flat_set_page_real(self.h, page_real) != 0:
            raise ValueError("flat_set_page_real")

    @property
    def idx_end(self) -> int:
        return int(self.L.flat_idx_end(self.h))

    @property
    def meta(self):
        """POINTER(HnswMetadata) — what the Postgres glue passes to hnsw_search."""
        return self.L.flat_meta(self.h)

    def add(self, vecs, labels=None) -> None:
        """append + hnsw_bind_point per row (embedding.c:606-701)."""
        vecs = _f32(vecs).reshape(-1, self.dim)
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint64)
            lp = _ptr(labels, _u64p)
        r = self.L.flat_add_many(self.h, _ptr(vecs, _f32p), lp, vecs.shape[0])
        if r < 0:
            raise RuntimeError(f"flat_add_many failed ({r})")

    def raw(self) -> np.ndarray:
        n = self.count * self.elem_size
        buf = (C.c_uint8 * n).from_address(self.L.flat_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8).copy()

    def raw_view(self) -> np.ndarray:
        n = self.count * self.elem_size
        buf = (C.c_uint8 * n).from_address(self.L.flat_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8)

    def load_raw(self, raw: np.ndarray, n: int) -> None:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.size == n * self.elem_size
        if self.L.flat_load_raw(self.h, raw.ctypes.data, n) != 0:
            raise MemoryError("flat_load_raw")

    def set_deleted(self, idx: int, deleted: bool = True) -> None:
        self.L.flat_set_deleted(self.h, idx, int(deleted))

    def search(self, q, efs: Optional[int] = None) -> np.ndarray:
        """hnsw_search through the C boundary: labels ascending by (dist, label)."""
        efs = efs or self.efs
        q = _f32(q)
        out = np.empty(efs, np.uint64)
        n = C.c_size_t(0)
        if self.L.flat_search(self.h, _ptr(q, _f32p), efs, _ptr(out, _u64p), C.byref(n)) != 0:
            raise RuntimeError("hnsw_search returned false")
        return out[:n.value].copy()

    def search_many(self, Q, efs: Optional[int] = None, nthreads: int = 1) -> dict:
        efs = efs or self.efs
        Q = _f32(Q).reshape(-1, self.dim)
        nq = Q.shape[0]
        lab = np.zeros((nq, efs), np.uint64)
        cnt = np.zeros(nq, np.uint32)
        ev = np.zeros(nq, np.uint32)
        hp = np.zeros(nq, np.uint32)
        sec = self.L.flat_search_many(self.h, _ptr(Q, _f32p), nq, efs, nthreads,
                                      _ptr(lab, _u64p), _ptr(cnt, _u32p), _ptr(ev, _u32p),
                                      _ptr(hp, _u32p))
        if sec < 0:
            raise RuntimeError("hnsw_search returned false")
        return dict(labels=lab, counts=cnt, evals=ev, hops=hp, seconds=sec)


_ref_lib = None


def _ref():
    global _ref_lib
    if _ref_lib is None:
        if not os.path.exists(_REF_SO):
            build_oracle()
        if not os.path.exists(_REF_SO):
            raise FileNotFoundError(
                f"{_REF_SO} missing: the reference-derived oracle can only be built where "
                "/root/reference exists (make -C oracle ref)")
        _ref_lib = _bind_flat(C.CDLL(_REF_SO))
    return _ref_lib


class RefIndex(_FlatIndexBase):
    """The unmodified reference hot path over the flat-memory host."""

    def __init__(self, *a, **k):
        self.L = _ref()
        super().__init__(*a, **k)


def ref_dist(func: int, a, b) -> float:
    a, b = _f32(a), _f32(b)
    return float(_ref().flat_dist(func, _ptr(a, _f32p), _ptr(b, _f32p), a.size))


def ref_dist_many(func: int, q, rows) -> np.ndarray:
    q, rows = _f32(q), _f32(rows)
    out = np.empty(rows.shape[0], np.float32)
    _ref().flat_dist_many(func, _ptr(q, _f32p), _ptr(rows, _f32p), rows.shape[0], q.size,
                          _ptr(out, _f32p))
    return out


_host_lib = None


def _host(shim_path: str):
    """Flat host whose hnsw_search/hnsw_bind_point/hnsw_dist_func resolve to the
    PRODUCT shim: load the shim first (global, lazy), then the host."""
    global _host_lib
    if _host_lib is None:
        if not os.path.exists(_HOST_SO):
            build_oracle()
        C.CDLL(shim_path, mode=C.RTLD_GLOBAL | os.RTLD_LAZY)
        _host_lib = _bind_flat(C.CDLL(_HOST_SO, mode=C.RTLD_GLOBAL | os.RTLD_NOW))
    return _host_lib


class FlatHostIndex(_FlatIndexBase):
    """Same host, but the hot path behind the boundary is libembedding_gpu.so —
    used by the drop-in tests (GPU only)."""

    def __init__(self, shim_path: str, *a, **k):
        self.L = _host(shim_path)
        super().__init__(*a, **k)


def lockstep_insert_compare(dim: int, m: int, efc: int, func: int, X, labels=None, tol: float = 1e-5) -> dict:
    """Insert-path parity of the canonical arithmetic against the COMPILED REFERENCE, insert by insert (hnswalg.cpp:117-232).

    Two serial builds of the same rows run side by side: `R` = oracle/_ref (the reference's own hnsw_bind_point) and `P` = the
    port in the canonical order with the reference's hnsw_dist_func as shadow arithmetic.  After every insert the lists the
    insert may have written (the new element's and those of both sides' selected neighbours) are compared; where they differ,
    P's lists are overwritten with R's, so that EVERY insert starts from the reference's own graph on both sides and each
    difference is one insert's own: it must come with a recorded decision (walk, heuristic pair test, candidate or list order)
    whose two values lie within `tol` relative in the canonical arithmetic — the north-star tolerance — and an insert without
    such a decision must have written the reference's bytes.  Returns the counts and the unexplained inserts (expected: none)."""
    X = _f32(X).reshape(-1, dim)
    n = X.shape[0]
    if labels is None:
        labels = np.arange(n, dtype=np.uint64)
    labels = np.ascontiguousarray(labels, dtype=np.uint64)
    R = RefIndex(dim, m, efc, 64, func, n)
    P = PortIndex(dim, m, efc, 64, func, n)
    P.shadow_reference_distances(True)
    esz, maxM = P.elem_size, 2 * m
    lw = maxM + 1
    differing, unexplained, margins, kinds = [], [], [], {}
    same_state_claims_broken = []
    for i in range(n):
        R.add(X[i:i + 1], labels[i:i + 1])
        _, kind, margin = P.add_shadowed(X[i], int(labels[i]))
        rv = R.raw_view().reshape(i + 1, esz)
        pv = P.raw_view().reshape(i + 1, esz)
        rl = rv[i, :lw * 4].view(np.uint32)
        pl = pv[i, :lw * 4].view(np.uint32)
        touched = {i} | set(rl[1:1 + rl[0]].tolist()) | set(pl[1:1 + pl[0]].tolist())
        bad = []
        for e in touched:
            a = rv[e, :lw * 4].view(np.uint32)
            b = pv[e, :lw * 4].view(np.uint32)
            if a[0] != b[0] or (a[1:1 + a[0]] != b[1:1 + a[0]]).any():
                bad.append(e)
        if bad:
            differing.append(i)
            kinds[kind] = kinds.get(kind, 0) + 1
            margins.append(margin)
            if kind == 0:
                same_state_claims_broken.append(i)
            elif not margin <= tol:
                unexplained.append((i, kind, margin))
            for e in touched:                                  # both sides go on from the reference's graph
                pv[e, :lw * 4] = rv[e, :lw * 4]
    return {"inserts": n, "inserts_with_differing_lists": len(differing), "first_differing": differing[:8],
            "decision_kinds": kinds, "largest_margin": max(margins) if margins else 0.0,
            "unexplained": unexplained, "no_decision_but_different": same_state_claims_broken}
