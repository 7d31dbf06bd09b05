"""Host-side face of the device mirror (include/hnsw_gpu.h) — thin ctypes plumbing.

Names follow the reference's domain: an *index* of *elements* (links | vector | label),
searched with a beam of ``efSearch``; see embedding.c:214-244 for the metadata derivation
mirrored by :func:`make_meta`.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from ._lib import HnswMetadata, check, gpu_lib

DIST_L2, DIST_COSINE, DIST_MANHATTAN = 0, 1, 2        # embedding.h:22-26
# opclass -> metric, embedding--0.3.6.sql:57-70
OPCLASS = {"ann_l2_ops": DIST_L2, "ann_cos_ops": DIST_COSINE, "ann_manhattan_ops": DIST_MANHATTAN}
DEFAULT_M, DEFAULT_EF_CONSTRUCTION, DEFAULT_EF_SEARCH = 100, 16, 64   # embedding.c:111-113
LABEL_DELETED = np.uint64(1) << np.uint64(48)          # embedding.c:44,948-953
NO_LABEL = np.uint64(0xFFFFFFFFFFFFFFFF)


def make_meta(dims: int, m: int = DEFAULT_M, efconstruction: int = DEFAULT_EF_CONSTRUCTION,
              efsearch: int = DEFAULT_EF_SEARCH, dist_func: int = DIST_L2) -> HnswMetadata:
    """The reloptions -> HnswMetadata derivation of hnsw_get_index (embedding.c:214-244)."""
    if dims <= 0:
        raise ValueError("HNSW index requires 'dims' to be specified")      # embedding.c:219-221
    mt = HnswMetadata()
    mt.dim = dims
    mt.M = m
    mt.maxM = 2 * m
    mt.data_size = dims * 4
    mt.offset_data = (mt.maxM + 1) * 4
    mt.offset_label = mt.offset_data + mt.data_size
    mt.size_data_per_element = mt.offset_label + 8
    mt.elems_per_page = (8192 - 24 - 4) // (mt.size_data_per_element + 4)
    if mt.elems_per_page == 0:
        raise ValueError("Element doesn't fit in Postgres page")            # embedding.c:230-231
    mt.efConstruction = efconstruction
    mt.efSearch = efsearch
    mt.dist_func = dist_func
    mt.enterpoint_node = 0
    return mt


def _torch():
    import torch
    return torch


def _dptr(t) -> int:
    return 0 if t is None else t.data_ptr()


class GpuIndex:
    """An HBM-resident mirror of one HNSW index on one MI355X."""

    def __init__(self, handle: int, meta: HnswMetadata, device: int):
        self._h = C.c_void_p(handle)
        self.meta = meta
        self.device = device
        self.L = gpu_lib()

    # ------------------------------------------------------------------ lifetime
    @classmethod
    def from_flat(cls, meta: HnswMetadata, elements: np.ndarray, n: int, device: int = 0) -> "GpuIndex":
        """Mirror `n` host element images ([count|links|vector|label], embedding.c:222-228)."""
        L = gpu_lib()
        elements = np.ascontiguousarray(elements, dtype=np.uint8)
        if elements.size != n * meta.size_data_per_element:
            raise ValueError("element image has the wrong size")
        h = C.c_void_p()
        check(L.hnsw_gpu_index_create_from_flat(C.byref(meta), elements.ctypes.data, n, device, C.byref(h)),
              "hnsw_gpu_index_create_from_flat")
        return cls(h.value, meta, device)

    @classmethod
    def empty(cls, meta: HnswMetadata, capacity: int, device: int = 0) -> "GpuIndex":
        L = gpu_lib()
        h = C.c_void_p()
        check(L.hnsw_gpu_index_create_empty(C.byref(meta), capacity, device, C.byref(h)),
              "hnsw_gpu_index_create_empty")
        return cls(h.value, meta, device)

    def close(self) -> None:
        if self._h:
            self.L.hnsw_gpu_index_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self) -> int:
        return self._h.value

    @property
    def count(self) -> int:
        return int(self.L.hnsw_gpu_index_count(self._h))

    # ------------------------------------------------------------------- content
    def append(self, vectors: np.ndarray, labels: Optional[np.ndarray] = None) -> None:
        vectors = np.ascontiguousarray(vectors, dtype=np.float32).reshape(-1, self.meta.dim)
        lp = None
        if labels is not None:
            labels = np.ascontiguousarray(labels, dtype=np.uint64)
            lp = labels.ctypes.data
        check(self.L.hnsw_gpu_index_append(self._h, vectors.ctypes.data, lp, vectors.shape[0]),
              "hnsw_gpu_index_append")

    def append_torch(self, vectors, labels=None) -> None:
        torch = _torch()
        assert vectors.is_cuda and vectors.dtype == torch.float32 and vectors.is_contiguous()
        s = torch.cuda.current_stream(vectors.device).cuda_stream
        check(self.L.hnsw_gpu_index_append_dev(self._h, vectors.data_ptr(), _dptr(labels),
                                               vectors.shape[0], s), "hnsw_gpu_index_append_dev")

    def link(self, first: int, count: int, max_batch: int = 0, ratio: int = 0, stream: int = 0) -> None:
        """hnsw_bind_point for the stored elements [first, first+count) (hnswalg.cpp:225-232,
        155-223, 117-153).  max_batch=1 == the reference's serial inserts, bit for bit."""
        check(self.L.hnsw_gpu_index_link(self._h, first, count, max_batch, ratio, stream), "hnsw_gpu_index_link")

    def reserve(self, capacity: int) -> None:
        """Room for `capacity` elements (a reallocation and a copy of the mirror when it has to grow)."""
        check(self.L.hnsw_gpu_index_reserve(self._h, int(capacity)), "hnsw_gpu_index_reserve")

    def insert_one(self, point: np.ndarray, label: int, candidates=None):
        """hnsw_bind_point's device side in one call (hnswalg.cpp:279-291, 225-232): the row becomes element `count` and is linked
        exactly as the reference's serial insert links it.  candidates = (element numbers, distances) of a base-mode walk with
        ef = efConstruction made for this point on this mirror (search_trace(..., base=True)): the insert then runs off that list
        (hnsw_gpu_index_insert_candidates).  Returns (mine, others): the new element's links and, per link, that neighbour's links."""
        p = np.ascontiguousarray(point, dtype=np.float32).reshape(self.meta.dim)
        maxM = int(self.meta.maxM)
        mine = np.zeros(maxM + 1, np.uint32)
        others = np.zeros(maxM * (maxM + 1), np.uint32)
        idx = self.count
        if candidates is None:
            check(self.L.hnsw_gpu_index_insert_one(self._h, p.ctypes.data, int(label), idx, mine.ctypes.data, others.ctypes.data),
                  "hnsw_gpu_index_insert_one")
        else:
            ci = np.ascontiguousarray(candidates[0], dtype=np.uint32)
            cd = np.ascontiguousarray(candidates[1], dtype=np.float32)
            check(self.L.hnsw_gpu_index_insert_candidates(self._h, p.ctypes.data, int(label), idx, ci.ctypes.data, cd.ctypes.data, len(ci),
                                                          mine.ctypes.data, others.ctypes.data), "hnsw_gpu_index_insert_candidates")
        k = int(mine[0])
        return mine[1:1 + k].copy(), [others[j * (maxM + 1) + 1:j * (maxM + 1) + 1 + int(others[j * (maxM + 1)])].copy() for j in range(k)]

    def insert_path_counts(self):
        """(inserts of this process through the two launches of csrc/device_insert.h, through the general builder path)"""
        out = (C.c_uint64 * 2)()
        self.L.hnsw_gpu_insert_path_counts(out)
        return int(out[0]), int(out[1])

    def update_from_flat(self, elements: np.ndarray, first: int, count: int) -> None:
        """Replace / add elements [first, first+count) from host element images."""
        elements = np.ascontiguousarray(elements, dtype=np.uint8)
        if elements.size != count * self.meta.size_data_per_element:
            raise ValueError("element image has the wrong size")
        check(self.L.hnsw_gpu_index_update_from_flat(self._h, elements.ctypes.data, first, count),
              "hnsw_gpu_index_update_from_flat")

    def export_flat(self) -> np.ndarray:
        out = np.empty(self.count * self.meta.size_data_per_element, np.uint8)
        check(self.L.hnsw_gpu_index_export_flat(self._h, out.ctypes.data), "hnsw_gpu_index_export_flat")
        return out

    def set_deleted(self, idx: int, deleted: bool = True) -> None:
        check(self.L.hnsw_gpu_index_set_deleted(self._h, idx, int(deleted)), "hnsw_gpu_index_set_deleted")

    def set_deleted_many(self, idx, deleted: bool = True) -> None:
        """Vacuum flags of many elements in one call (hnsw_gpu_index_set_deleted_batch)."""
        a = np.ascontiguousarray(idx, dtype=np.uint32)
        check(self.L.hnsw_gpu_index_set_deleted_batch(self._h, a.ctypes.data_as(C.POINTER(C.c_uint32)), a.size, int(deleted)),
              "hnsw_gpu_index_set_deleted_batch")

    # -------------------------------------------------------------------- search
    def search(self, queries: np.ndarray, ef: Optional[int] = None):
        """Batch of hnsw_search() calls with host buffers.
        Returns (labels[nq, ef] u64, dists[nq, ef] f32, counts[nq] u32)."""
        ef = int(ef or self.meta.efSearch)
        queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, self.meta.dim)
        nq = queries.shape[0]
        labels = np.empty((nq, ef), np.uint64)
        dists = np.empty((nq, ef), np.float32)
        counts = np.empty(nq, np.uint32)
        check(self.L.hnsw_gpu_search_batch(self._h, queries.ctypes.data, nq, ef, labels.ctypes.data,
                                           dists.ctypes.data, counts.ctypes.data), "hnsw_gpu_search_batch")
        return labels, dists, counts

    def search_trace(self, query: np.ndarray, ef: Optional[int] = None, base: bool = False, pops_cap: int = 1 << 16):
        """One query with its walk (hnsw_gpu_search_trace): (labels-or-element-numbers[count] u64, dists[count] f32,
        pops[npops] u32 = the elements the walk expanded in order, evals)."""
        ef = int(ef or self.meta.efSearch)
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.meta.dim)
        labels = np.empty(ef, np.uint64)
        dists = np.empty(ef, np.float32)
        pops = np.empty(pops_cap, np.uint32)
        cnt, npops, nev = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(self.L.hnsw_gpu_search_trace(self._h, q.ctypes.data, ef, int(base), labels.ctypes.data, dists.ctypes.data,
                                           C.byref(cnt), pops.ctypes.data_as(C.POINTER(C.c_uint32)), pops_cap,
                                           C.byref(npops), C.byref(nev)), "hnsw_gpu_search_trace")
        return labels[:cnt.value], dists[:cnt.value], pops[:min(npops.value, pops_cap)], int(nev.value)

    def search_trace_polled(self, query: np.ndarray, ef: Optional[int] = None, base: bool = False, pops_cap: int = 1 << 14,
                            slice_: int = 5):
        """The same through hnsw_gpu_search_trace_begin / _poll / _end: the pops are taken while the kernel runs, at most
        `slice_` per poll.  Returns (labels, dists, pops, evals, polls)."""
        ef = int(ef or self.meta.efSearch)
        q = np.ascontiguousarray(query, dtype=np.float32).reshape(self.meta.dim)
        check(self.L.hnsw_gpu_search_trace_begin(self._h, q.ctypes.data, ef, int(base), pops_cap), "hnsw_gpu_search_trace_begin")
        pops = np.empty(pops_cap + slice_, np.uint32)
        have, polls = 0, 0
        got, fin = C.c_size_t(0), C.c_int(0)
        while True:
            check(self.L.hnsw_gpu_search_trace_poll(self._h, pops[have:].ctypes.data_as(C.POINTER(C.c_uint32)),
                                                    min(slice_, pops_cap - have), C.byref(got), C.byref(fin)), "hnsw_gpu_search_trace_poll")
            have += got.value
            polls += 1
            if fin.value:
                break
        labels = np.empty(ef, np.uint64)
        dists = np.empty(ef, np.float32)
        cnt, npops, nev = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        check(self.L.hnsw_gpu_search_trace_end(self._h, labels.ctypes.data, dists.ctypes.data, C.byref(cnt), C.byref(npops), C.byref(nev)),
              "hnsw_gpu_search_trace_end")
        assert have == min(npops.value, pops_cap)
        return labels[:cnt.value], dists[:cnt.value], pops[:have].copy(), int(nev.value), polls

    def search_torch(self, queries, ef: Optional[int] = None, out=None, stats: bool = False, base: bool = False):
        """Same with everything resident in HBM (torch tensors only carry the pointers).
        `out` may be a dict from a previous call to reuse its buffers.  With base=True runs
        searchBaseLayer only and returns element numbers under 'idx'."""
        torch = _torch()
        ef = int(ef or self.meta.efSearch)
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq = queries.shape[0]
        dev = queries.device
        if out is None:
            out = {}
            if base:
                out["idx"] = torch.empty((nq, ef), dtype=torch.int32, device=dev)
            else:
                out["labels"] = torch.empty((nq, ef), dtype=torch.int64, device=dev)
            out["dists"] = torch.empty((nq, ef), dtype=torch.float32, device=dev)
            out["counts"] = torch.empty(nq, dtype=torch.int32, device=dev)
            out["stats"] = torch.zeros((nq, 2), dtype=torch.int32, device=dev) if stats else None
        s = torch.cuda.current_stream(dev).cuda_stream
        if base:
            check(self.L.hnsw_gpu_search_base_dev(self._h, queries.data_ptr(), nq, ef, out["idx"].data_ptr(),
                                                  out["dists"].data_ptr(), out["counts"].data_ptr(),
                                                  _dptr(out.get("stats")), s), "hnsw_gpu_search_base_dev")
        else:
            check(self.L.hnsw_gpu_search_batch_dev(self._h, queries.data_ptr(), nq, ef, out["labels"].data_ptr(),
                                                   out["dists"].data_ptr(), out["counts"].data_ptr(),
                                                   _dptr(out.get("stats")), s), "hnsw_gpu_search_batch_dev")
        return out

    def last_search_ms(self, back: int = 0) -> float:
        """Device time of the search kernel launched `back` launches ago (HIP events recorded on
        the launch stream around the kernel; the last 64 launches are kept)."""
        ms = C.c_float(0)
        check(self.L.hnsw_gpu_search_ms(self._h, back, C.byref(ms)), "hnsw_gpu_search_ms")
        return float(ms.value)

    def last_batch_ms(self):
        """(upload ms, kernel ms, download ms) of the last host-pointer search of more than 16 queries (hnsw_gpu_last_batch_ms)."""
        v = (C.c_float * 3)()
        check(self.L.hnsw_gpu_last_batch_ms(self._h, v), "hnsw_gpu_last_batch_ms")
        return float(v[0]), float(v[1]), float(v[2])

    def last_search_kernel(self) -> str:
        """Symbol of the kernel the last search launch ran, as rocprofv3 prints it."""
        buf = C.create_string_buffer(128)
        check(self.L.hnsw_gpu_last_search_kernel(self._h, buf, 128), "hnsw_gpu_last_search_kernel")
        return buf.value.decode()

    def team_counters(self):
        """Launch-wide counters of the team form (needs HNSW_GPU_TEAM_COUNTERS=1): dict."""
        v = (C.c_uint32 * 16)()
        check(self.L.hnsw_gpu_team_counters(self._h, v), "hnsw_gpu_team_counters")
        names = ("hops_with_helpers", "link_hits", "ids_looked_up", "dist_hits", "hops_that_scored", "hops", "wait_polls",
                 "cyc_pop_links", "cyc_dists", "cyc_accept", "hops_that_waited", "helper_elements", "helper_cycles")
        return {k: int(v[i]) for i, k in enumerate(names)}

    def debug_counters(self):
        """The 16 raw launch-wide debug counters (HNSW_GPU_TEAM_COUNTERS=1; -DHNSW_HOP_STAMPS builds put the per-section
        cycle sums of the walking waves there: hops, then pop / link wait / visited / scoring / accept / walk / emit in
        units of 64 cycles)."""
        v = (C.c_uint32 * 16)()
        check(self.L.hnsw_gpu_team_counters(self._h, v), "hnsw_gpu_team_counters")
        return [int(x) for x in v]

    def gather_roof(self, loads_per_lane: int = 12, waves_per_cu: int = 16, iters: int = 200) -> float:
        """GB/s of a dependency-free random gather of whole rows of THIS mirror's row table — the practical
        roof of the search kernel's access pattern (csrc/device_roof.h)."""
        v = C.c_float(0)
        check(self.L.hnsw_gpu_gather_roof(self._h, loads_per_lane, waves_per_cu, iters, C.byref(v)), "hnsw_gpu_gather_roof")
        return float(v.value)

    def search_traced_torch(self, queries, ef: int, evals_cap: int = 4096):
        """hnsw_gpu_search_traced_dev: search_torch(..., stats=True) plus out["evals"] (nq x evals_cap int32: the rows each walk
        scored, in order; out["stats"][:, 0] of them are valid) and out["times"] (nq x 2 int64: 100 MHz device clock at the
        start of the query and at the end of its walk).  Measurement only."""
        torch = _torch()
        nq = queries.shape[0]
        dev = queries.device
        out = {"labels": torch.empty((nq, ef), dtype=torch.int64, device=dev), "dists": torch.empty((nq, ef), dtype=torch.float32, device=dev),
               "counts": torch.empty((nq,), dtype=torch.int32, device=dev), "stats": torch.empty((nq, 2), dtype=torch.int32, device=dev),
               "evals": torch.empty((nq, evals_cap), dtype=torch.int32, device=dev), "times": torch.zeros((nq, 2), dtype=torch.int64, device=dev)}
        s = torch.cuda.current_stream(dev).cuda_stream
        check(self.L.hnsw_gpu_search_traced_dev(self._h, queries.data_ptr(), nq, ef, out["labels"].data_ptr(), out["dists"].data_ptr(),
                                                out["counts"].data_ptr(), out["stats"].data_ptr(), out["evals"].data_ptr(), evals_cap,
                                                out["times"].data_ptr(), s), "hnsw_gpu_search_traced_dev")
        return out

    def replay_roof(self, traced: dict, slots: int, kb: int = 12, rpg: int = 2, word_sum: bool = False, parts: int = 1):
        """(ms, bytes) of hnsw_gpu_replay_roof over the trace of a search_traced_torch launch with load shape <kb, rpg>; with
        word_sum=True also the sum mod 2^64 of the bit patterns of every word the replay read for the trace (tests); parts > 1:
        every query's trace is cut into that many pieces gathered by different waves (hnsw_gpu_replay_roof_parts)."""
        ms, by, ws = C.c_float(0), C.c_double(0), C.c_uint64(0)
        ev = traced["evals"]
        check(self.L.hnsw_gpu_replay_roof_parts(self._h, ev.data_ptr(), ev.shape[1], traced["stats"].data_ptr(), ev.shape[0], slots, kb, rpg,
                                                parts, C.byref(ms), C.byref(by), C.byref(ws) if word_sum else None), "hnsw_gpu_replay_roof")
        return (float(ms.value), float(by.value), int(ws.value)) if word_sum else (float(ms.value), float(by.value))

    def health(self) -> dict:
        """Health words of the default search workspace (include/hnsw_gpu.h, hnsw_gpu_index_health): all zero in a healthy life."""
        v = (C.c_uint32 * 8)()
        check(self.L.hnsw_gpu_index_health(self._h, v), "hnsw_gpu_index_health")
        return {"abort_pending": int(v[0]), "slice_timeouts": int(v[1]), "package_timeouts": int(v[2]), "aborted_waves": int(v[3]),
                "slices_delivered": int(v[4]), "abort_requests": int(v[5])}

    def abort(self) -> None:
        """Ask the search launches of this mirror that are in flight to end (callable from any thread)."""
        check(self.L.hnsw_gpu_index_abort(self._h), "hnsw_gpu_index_abort")

    def last_search_clock_mhz(self) -> float:
        """Shader clock (MHz) the last search launch ran at, measured by its first wave (include/hnsw_gpu_diag.h); 0.0 = not recorded."""
        v = C.c_double(0.0)
        check(self.L.hnsw_gpu_last_search_clock_mhz(self._h, C.byref(v)), "hnsw_gpu_last_search_clock_mhz")
        return float(v.value)

    def placement(self) -> dict:
        """Device address and size of the mirror's arena, its three arrays and the default search workspace (include/hnsw_gpu_diag.h):
        {"rows": (address, bytes), ...} plus "aligned_2MiB": do the three arrays start on 2 MiB boundaries."""
        v = (C.c_uint64 * 16)()
        check(self.L.hnsw_gpu_index_placement(self._h, v), "hnsw_gpu_index_placement")
        names = ("arena", "rows", "links", "labels", "visited_bitmaps", "bitmap_logs", "prune_scratch", "ticket")
        out = {k: (int(v[2 * i]), int(v[2 * i + 1])) for i, k in enumerate(names)}
        out["aligned_2MiB"] = all(out[k][0] % (2 << 20) == 0 for k in ("rows", "links", "labels"))
        return out

    def last_search_slots(self) -> int:
        v = C.c_uint32(0)
        check(self.L.hnsw_gpu_last_search_slots(self._h, C.byref(v)), "hnsw_gpu_last_search_slots")
        return int(v.value)

    def bruteforce_torch(self, queries, k: int, mfma: bool = False):
        """Exact k nearest elements (idx, dists) by exhaustive scoring — recall ground truth.
        mfma=True runs the Q x N part as an f32 GEMM on the matrix cores (same result)."""
        torch = _torch()
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq = queries.shape[0]
        idx = torch.empty((nq, k), dtype=torch.int32, device=queries.device)
        dst = torch.empty((nq, k), dtype=torch.float32, device=queries.device)
        s = torch.cuda.current_stream(queries.device).cuda_stream
        step = 4096 if mfma else 32768
        for q0 in range(0, nq, step):
            q1 = min(nq, q0 + step)
            fn = self.L.hnsw_gpu_bruteforce_mfma_dev if mfma else self.L.hnsw_gpu_bruteforce_dev
            check(fn(self._h, queries[q0:q1].data_ptr(), q1 - q0, k,
                     idx[q0:q1].data_ptr(), dst[q0:q1].data_ptr(), s), "hnsw_gpu_bruteforce")
        return idx, dst


class SearchContext:
    """An extra search workspace on a mirror: batches launched through different contexts on
    different torch streams overlap on the GPU (include/hnsw_gpu.h, "Search contexts")."""

    def __init__(self, index: GpuIndex):
        self.index = index
        self.L = index.L
        h = C.c_void_p()
        check(self.L.hnsw_gpu_ctx_create(index._h, C.byref(h)), "hnsw_gpu_ctx_create")
        self._h = h

    def close(self) -> None:
        if self._h:
            self.L.hnsw_gpu_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_walkers(self, per_block: int) -> None:
        """Walking waves per block of this context's small (team) launches: hnsw_gpu_ctx_set_walkers; 0 = by launch size."""
        check(self.L.hnsw_gpu_ctx_set_walkers(self._h, int(per_block)), "hnsw_gpu_ctx_set_walkers")

    def search_torch(self, queries, ef: int, out: dict, stream=None):
        """Enqueue one batch on `stream` (a torch.cuda.Stream; default: the current one)."""
        torch = _torch()
        s = (stream or torch.cuda.current_stream(queries.device)).cuda_stream
        check(self.L.hnsw_gpu_search_batch_ctx(self._h, queries.data_ptr(), queries.shape[0], ef,
                                               out["labels"].data_ptr(), out["dists"].data_ptr(),
                                               out["counts"].data_ptr(), _dptr(out.get("stats")), s),
              "hnsw_gpu_search_batch_ctx")
        return out

    def last_search_ms(self, back: int = 0) -> float:
        """Device milliseconds of this context's search launch `back` launches ago (HIP events on its stream; waits for it)."""
        v = C.c_float(0)
        check(self.L.hnsw_gpu_ctx_search_ms(self._h, back, C.byref(v)), "hnsw_gpu_ctx_search_ms")
        return float(v.value)

    def search_host(self, Q: np.ndarray, ef: int):
        """Host arrays in and out on the context's own stream (hnsw_gpu_search_batch_ctx_host)."""
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, int(self.index.meta.dim))
        nq = Q.shape[0]
        lab = np.empty((nq, ef), np.uint64)
        dst = np.empty((nq, ef), np.float32)
        cnt = np.empty(nq, np.uint32)
        check(self.L.hnsw_gpu_search_batch_ctx_host(self._h, Q.ctypes.data, nq, ef, lab.ctypes.data, dst.ctypes.data,
                                                    cnt.ctypes.data), "hnsw_gpu_search_batch_ctx_host")
        return lab, dst, cnt

    def search_streamed(self, Q: np.ndarray, ef: int, timeout: float = 60.0):
        """Streamed completion (hnsw_gpu_search_batch_ctx_flags): queries, results and per-query
        completion flags live in pinned host memory the kernel reads and writes directly.  Returns
        (labels, dists, counts, order) where `order` lists the queries in the order their flags
        were seen — what a server uses to answer each caller as soon as its walk has ended."""
        import time
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, int(self.index.meta.dim))
        nq, dim = Q.shape
        sizes = [nq * dim * 4, nq * ef * 8, nq * ef * 4, nq * 4, nq * 4]
        offs = np.cumsum([0] + [(b + 255) // 256 * 256 for b in sizes])
        base = self.L.hnsw_gpu_host_alloc(int(offs[-1]))
        if not base:
            raise MemoryError("hnsw_gpu_host_alloc")
        try:
            view = lambda i, dt, shape: np.frombuffer((C.c_uint8 * sizes[i]).from_address(base + int(offs[i])), dtype=dt).reshape(shape)
            q, lab, dst = view(0, np.float32, (nq, dim)), view(1, np.uint64, (nq, ef)), view(2, np.float32, (nq, ef))
            cnt, flg = view(3, np.uint32, (nq,)), view(4, np.uint32, (nq,))
            q[:] = Q
            flg[:] = 0
            check(self.L.hnsw_gpu_search_batch_ctx_flags(self._h, q.ctypes.data, nq, ef, lab.ctypes.data, dst.ctypes.data,
                                                         cnt.ctypes.data, None, flg.ctypes.data),
                  "hnsw_gpu_search_batch_ctx_flags")
            order, seen = [], np.zeros(nq, bool)
            t_end = time.time() + timeout
            while len(order) < nq:
                new = np.flatnonzero((flg != 0) & ~seen)
                seen[new] = True
                order.extend(new.tolist())
                if time.time() > t_end:
                    raise TimeoutError("completion flags did not arrive")
            while self.L.hnsw_gpu_ctx_idle(self._h) == 0:
                if time.time() > t_end:
                    raise TimeoutError("launch did not retire")
            return lab.copy(), dst.copy(), cnt.copy(), np.array(order)
        finally:
            self.L.hnsw_gpu_host_free(base)


# ---------------------------------------------------------------------- distances
class SearchStream:
    """A resident search launch fed from the host (include/hnsw_gpu.h, "Streams"): numpy views of the pinned ring + publish / poll.
    `submit(Q)` writes queries into the next slots and publishes them, returning their slot numbers; `wait(slots)` spins on their
    completion flags and returns (labels, dists, counts) rows.  One producer thread."""

    def __init__(self, ctx: "SearchContext", ef: int, ring: int = 4096, walkers: int = 0):
        self.ctx, self.L, self.ef, self.ring = ctx, ctx.L, ef, ring
        h = C.c_void_p()
        check(self.L.hnsw_gpu_stream_open(ctx._h, ef, ring, walkers, C.byref(h)), "hnsw_gpu_stream_open")
        self._h = h
        ptr = [C.c_void_p() for _ in range(5)]
        check(self.L.hnsw_gpu_stream_buffers(h, *[C.byref(p) for p in ptr]), "hnsw_gpu_stream_buffers")
        dim = int(ctx.index.meta.dim)

        def view(p, ctype, shape):
            n = int(np.prod(shape))
            return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), shape=(n,)).reshape(shape)
        self.q = view(ptr[0], C.c_float, (ring, dim))
        self.labels = view(ptr[1], C.c_uint64, (ring, ef))
        self.dists = view(ptr[2], C.c_float, (ring, ef))
        self.counts = view(ptr[3], C.c_uint32, (ring,))
        self.flags = view(ptr[4], C.c_uint32, (ring,))
        self.published = 0

    def submit(self, Q: np.ndarray) -> np.ndarray:
        Q = np.ascontiguousarray(Q, dtype=np.float32).reshape(-1, self.q.shape[1])
        slots = (self.published + np.arange(Q.shape[0], dtype=np.int64)) & (self.ring - 1)
        self.q[slots] = Q
        self.flags[slots] = 0
        self.published = (self.published + Q.shape[0]) & 0xFFFFFFFF
        check(self.L.hnsw_gpu_stream_publish(self._h, self.published), "hnsw_gpu_stream_publish")
        return slots

    def wait(self, slots: np.ndarray, timeout: float = 30.0):
        import time as _t
        t0 = _t.time()
        while not self.flags[slots].all():
            if _t.time() - t0 > timeout:
                raise TimeoutError(f"{int((self.flags[slots] == 0).sum())} of {len(slots)} stream queries unanswered after {timeout} s "
                                   f"(launch alive: {self.alive()})")
        return self.labels[slots].copy(), self.dists[slots].copy(), self.counts[slots].copy()

    def alive(self) -> bool:
        return self.L.hnsw_gpu_stream_alive(self._h) == 1

    def close(self) -> None:
        if self._h:
            h, self._h = self._h, C.c_void_p()
            self.q = self.labels = self.dists = self.counts = self.flags = None
            check(self.L.hnsw_gpu_stream_close(h), "hnsw_gpu_stream_close")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dist_batch(func: int, q: np.ndarray, rows: np.ndarray) -> np.ndarray:
    """out[i] = hnsw_dist_func(func, q, rows[i]) on the device (distfunc.c:171-174)."""
    L = gpu_lib()
    q = np.ascontiguousarray(q, dtype=np.float32).ravel()
    rows = np.ascontiguousarray(rows, dtype=np.float32).reshape(-1, q.size)
    out = np.empty(rows.shape[0], np.float32)
    check(L.hnsw_gpu_dist_batch(func, q.ctypes.data, rows.ctypes.data, rows.shape[0], q.size, out.ctypes.data),
          "hnsw_gpu_dist_batch")
    return out


def _scalar(func: int, a, b) -> float:
    a = np.ascontiguousarray(a, dtype=np.float32).ravel()
    b = np.ascontiguousarray(b, dtype=np.float32).ravel()
    if a.size != b.size:                      # calc_distance, embedding.c:1030-1035
        raise ValueError(f"Different array dimensions {a.size} and {b.size}")
    return float(dist_batch(func, a, b[None, :])[0])


def l2_distance(a, b) -> float:
    """SQL l2_distance(real[], real[]) / operator <-> (embedding--0.3.6.sql:20-21,31-35)."""
    return _scalar(DIST_L2, a, b)


def cosine_distance(a, b) -> float:
    """SQL cosine_distance / operator <=> (embedding--0.3.6.sql:23-24,36-40)."""
    return _scalar(DIST_COSINE, a, b)


def manhattan_distance(a, b) -> float:
    """SQL manhattan_distance / operator <~> (embedding--0.3.6.sql:26-27,41-44)."""
    return _scalar(DIST_MANHATTAN, a, b)


class LocalShardedIndex:
    """A row-sharded index inside one process (include/hnsw_gpu.h, hnsw_gpu_sharded_*): shards on one or
    several devices, per-shard search + one device merge, no torch.distributed involved.  The shards are
    GpuIndex objects the caller built (labels globally unique) and stay owned by the caller."""

    def __init__(self, shards):
        self.shards = list(shards)
        self.L = gpu_lib()
        arr = (C.c_void_p * len(self.shards))(*[sh.handle for sh in self.shards])
        h = C.c_void_p()
        check(self.L.hnsw_gpu_sharded_create(arr, len(self.shards), C.byref(h)), "hnsw_gpu_sharded_create")
        self._h = h

    @classmethod
    def build(cls, rows, meta, nshards: int, devices=None, max_batch: int = 0, ratio: int = 0):
        """Split host rows [n, dim] into `nshards` contiguous ranges, one graph per range on
        devices[i % len(devices)]; label = global row number."""
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        n = rows.shape[0]
        ndev = gpu_lib().hnsw_gpu_device_count()
        devices = list(devices) if devices else list(range(max(1, min(ndev, nshards))))
        shards = []
        for i in range(nshards):
            lo, hi = n * i // nshards, n * (i + 1) // nshards
            ix = GpuIndex.empty(meta, max(hi - lo, 1), device=devices[i % len(devices)])
            ix.append(rows[lo:hi], np.arange(lo, hi, dtype=np.uint64))
            ix.link(0, hi - lo, max_batch, ratio)
            shards.append(ix)
        return cls(shards)

    def close(self) -> None:
        if self._h:
            self.L.hnsw_gpu_sharded_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def search(self, queries: np.ndarray, ef: int):
        """Host arrays: (labels[nq, ef] u64, dists[nq, ef] f32, counts[nq] u32) of the merged result."""
        dim = int(self.shards[0].meta.dim)
        queries = np.ascontiguousarray(queries, dtype=np.float32).reshape(-1, dim)
        nq = queries.shape[0]
        labels = np.empty((nq, ef), np.uint64)
        dists = np.empty((nq, ef), np.float32)
        counts = np.empty(nq, np.uint32)
        check(self.L.hnsw_gpu_sharded_search(self._h, queries.ctypes.data, nq, ef, labels.ctypes.data,
                                             dists.ctypes.data, counts.ctypes.data), "hnsw_gpu_sharded_search")
        return labels, dists, counts

    def search_torch(self, queries, ef: int):
        """Device tensors on the device of shard 0; enqueued on the current torch stream."""
        torch = _torch()
        assert queries.is_cuda and queries.dtype == torch.float32 and queries.is_contiguous()
        nq, dev = queries.shape[0], queries.device
        ol = torch.empty((nq, ef), dtype=torch.int64, device=dev)
        od = torch.empty((nq, ef), dtype=torch.float32, device=dev)
        oc = torch.empty(nq, dtype=torch.int32, device=dev)
        s = torch.cuda.current_stream(dev).cuda_stream
        check(self.L.hnsw_gpu_sharded_search_dev(self._h, queries.data_ptr(), nq, ef, ol.data_ptr(), od.data_ptr(),
                                                 oc.data_ptr(), s), "hnsw_gpu_sharded_search_dev")
        return ol, od, oc

    def last_ms(self) -> dict:
        """Where the last search's time went (hnsw_gpu_sharded_last_ms): per shard the search kernel and what followed it until its
        lists were in the merge device's buffer, the merge kernel, and whether a shard stores into the merge device directly."""
        n = len(self.shards)
        sm, pm, mm, dr = (C.c_float * n)(), (C.c_float * n)(), C.c_float(0), (C.c_int * n)()
        check(self.L.hnsw_gpu_sharded_last_ms(self._h, sm, pm, C.byref(mm), dr), "hnsw_gpu_sharded_last_ms")
        return {"search_ms": [float(x) for x in sm], "peer_ms": [float(x) for x in pm], "merge_ms": float(mm.value),
                "direct_peer_stores": [bool(x) for x in dr], "devices": [int(sh.device) for sh in self.shards]}


def merge_packed_torch(blocks, nq: int, ef: int, out=None):
    """Merge the gathered per-rank blocks of ShardedIndex ([world, block_bytes] uint8, block =
    [labels nq*ef*8 | dists nq*ef*4 | pad]) in place: the strided merge entry reads every list where the
    all-gather put it.  out: (labels[nq, ef] int64, dists[nq, ef] float32, counts[nq] int32) to write into (allocated otherwise)."""
    torch = _torch()
    L = gpu_lib()
    assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
    world, block = blocks.shape
    assert block % 8 == 0 and block >= nq * ef * 12
    dev = blocks.device
    if out is not None:
        ol, od, oc = out
    else:
        ol = torch.empty((nq, ef), dtype=torch.int64, device=dev)
        od = torch.empty((nq, ef), dtype=torch.float32, device=dev)
        oc = torch.empty(nq, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    base = blocks.data_ptr()
    check(L.hnsw_gpu_merge_topk_strided_dev(dev.index or 0, base, block // 8, base + nq * ef * 8, block // 4, world, nq, ef,
                                            ol.data_ptr(), od.data_ptr(), oc.data_ptr(), s), "hnsw_gpu_merge_topk_strided_dev")
    return ol, od, oc


def merge_topk_torch(labels, dists, ef: int):
    """Merge per-shard result lists [nlists, nq, ef] into the ef best per query (device)."""
    torch = _torch()
    L = gpu_lib()
    assert labels.is_cuda and labels.dtype == torch.int64 and labels.is_contiguous()
    assert dists.is_cuda and dists.dtype == torch.float32 and dists.is_contiguous()
    nlists, nq = labels.shape[0], labels.shape[1]
    dev = labels.device
    ol = torch.empty((nq, ef), dtype=torch.int64, device=dev)
    od = torch.empty((nq, ef), dtype=torch.float32, device=dev)
    oc = torch.empty(nq, dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev).cuda_stream
    check(L.hnsw_gpu_merge_topk_dev(dev.index or 0, labels.data_ptr(), dists.data_ptr(), nlists, nq, ef,
                                    ol.data_ptr(), od.data_ptr(), oc.data_ptr(), s), "hnsw_gpu_merge_topk_dev")
    return ol, od, oc
