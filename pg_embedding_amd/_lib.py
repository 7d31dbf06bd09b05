"""ctypes loader for the in-tree gfx950 libraries.  Fails loudly: there is no CPU path."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)


class HnswMetadata(C.Structure):
    """ctypes image of HnswMetadata (embedding.h:28-42 == include/hnsw_abi.h)."""
    _fields_ = [(n, C.c_size_t) for n in (
        "dim", "data_size", "offset_data", "offset_label", "size_data_per_element",
        "elems_per_page", "M", "maxM", "efConstruction", "efSearch")] + [
        ("enterpoint_node", C.c_uint32), ("dist_func", C.c_int)]


class LibraryMissing(RuntimeError):
    pass


_gpu = None
_shim = None


def _env_sync_on() -> bool:
    return os.environ.get("PGEMB_ENV_SYNC") == "1"


class _EnvSyncLib:
    """PGEMB_ENV_SYNC=1 (test tiers, experiment scripts): the library handle forwards HNSW_GPU_* changes of os.environ to the
    library (hnsw_gpu_config_set) before every call that launches a search, an insert or a build — see sync_env below."""
    _LAUNCH = ("hnsw_gpu_search", "hnsw_gpu_index_insert", "hnsw_gpu_index_link", "hnsw_gpu_sharded_", "hnsw_gpu_index_create")

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        if name.startswith(self._LAUNCH):
            sync_env()
        return getattr(self._lib, name)

    def __setattr__(self, name, value):
        setattr(self._lib, name, value)


def _preload_torch_runtime() -> None:
    # PyTorch-ROCm ships its own libamdhip64.so.7; whichever copy is loaded first wins
    # for the whole process.  When torch is going to be used (device tensors, streams,
    # torch.distributed) it must be the one that is loaded first.
    if os.environ.get("PGEMB_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import torch  # noqa: F401
    except Exception:
        pass


def _build_once() -> None:
    """The libraries are normally prebuilt by __graft_entry__.build(); if they are absent, compile
    them here — under a file lock, because one process per GPU may get here at the same time."""
    import fcntl
    os.makedirs(_build.LIBDIR, exist_ok=True)
    with open(os.path.join(_build.LIBDIR, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if not os.path.exists(_build.GPU_LIB) or not os.path.exists(_build.SHIM_LIB):
                try:
                    _build.build()
                except Exception as e:      # no hipcc: report through LibraryMissing below
                    raise LibraryMissing(f"cannot build the gfx950 libraries: {e}") from e
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def gpu_lib():
    """libhnsw_gpu.so with argtypes set (include/hnsw_gpu.h)."""
    global _gpu
    if _gpu is not None:
        return _gpu
    if not os.path.exists(_build.GPU_LIB):
        _build_once()
    if not os.path.exists(_build.GPU_LIB):
        raise LibraryMissing(
            f"{_build.GPU_LIB} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  pg_embedding_amd has no CPU fallback.")
    _preload_torch_runtime()
    # PGEMB_GPU_LIB: another build of the same C-ABI instead of the product library — an experiment build for A/B runs of kernel
    # variants (build.py variant ...), or the test tier's SIMT-emulated build of the kernel source (tests/emu).  Never silent:
    # whoever reads the output of such a run sees which library it used.
    override = os.environ.get("PGEMB_GPU_LIB")
    if override:
        import sys
        print(f"pg_embedding_amd: PGEMB_GPU_LIB is set: using {override} instead of the product library", file=sys.stderr)
    L = C.CDLL(override or _build.GPU_LIB, mode=C.RTLD_GLOBAL)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    MP = C.POINTER(HnswMetadata)
    L.hnsw_gpu_last_error.restype = C.c_char_p
    L.hnsw_gpu_device_count.restype = i32
    L.hnsw_gpu_index_create_from_flat.argtypes = [MP, vp, sz, i32, C.POINTER(vp)]
    L.hnsw_gpu_index_create_empty.argtypes = [MP, sz, i32, C.POINTER(vp)]
    L.hnsw_gpu_index_append.argtypes = [vp, vp, vp, sz]
    L.hnsw_gpu_index_append_dev.argtypes = [vp, vp, vp, sz, vp]
    L.hnsw_gpu_index_export_flat.argtypes = [vp, vp]
    L.hnsw_gpu_index_link.argtypes = [vp, sz, sz, sz, sz, vp]
    L.hnsw_gpu_index_reserve.argtypes = [vp, sz]
    L.hnsw_gpu_index_update_from_flat.argtypes = [vp, vp, sz, sz]
    L.hnsw_gpu_index_get_links.argtypes = [vp, C.c_uint32, vp]
    L.hnsw_gpu_index_set_deleted.argtypes = [vp, C.c_uint32, i32]
    L.hnsw_gpu_index_set_deleted_batch.argtypes = [vp, _u32p, sz, i32]
    L.hnsw_gpu_search_trace.argtypes = [vp, vp, sz, i32, vp, vp, _u32p, _u32p, sz, _u32p, _u32p]
    L.hnsw_gpu_search_trace_begin.argtypes = [vp, vp, sz, i32, sz]
    L.hnsw_gpu_search_trace_poll.argtypes = [vp, _u32p, sz, C.POINTER(sz), C.POINTER(i32)]
    L.hnsw_gpu_search_trace_end.argtypes = [vp, vp, vp, _u32p, _u32p, _u32p]
    L.hnsw_gpu_index_count.restype = sz
    L.hnsw_gpu_index_count.argtypes = [vp]
    L.hnsw_gpu_index_device.argtypes = [vp]
    L.hnsw_gpu_index_destroy.restype = None
    L.hnsw_gpu_index_destroy.argtypes = [vp]
    L.hnsw_gpu_search_batch.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.hnsw_gpu_search_batch_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hnsw_gpu_search_base_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hnsw_gpu_last_search_ms.argtypes = [vp, _f32p]
    L.hnsw_gpu_search_ms.argtypes = [vp, C.c_uint, _f32p]
    L.hnsw_gpu_last_search_slots.argtypes = [vp, _u32p]
    L.hnsw_gpu_index_health.argtypes = [vp, _u32p]
    L.hnsw_gpu_index_insert_one.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp]
    L.hnsw_gpu_index_insert_candidates.argtypes = [vp, vp, C.c_uint64, C.c_uint32, vp, vp, C.c_uint32, vp, vp]
    L.hnsw_gpu_insert_path_counts.argtypes = [vp]
    L.hnsw_gpu_insert_path_counts.restype = None
    L.hnsw_gpu_index_capacity.restype = sz
    L.hnsw_gpu_index_capacity.argtypes = [vp]
    L.hnsw_gpu_search_traced_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp, sz, vp, vp]
    L.hnsw_gpu_replay_roof.argtypes = [vp, vp, sz, vp, sz, C.c_uint, i32, i32, _f32p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.hnsw_gpu_replay_roof_parts.argtypes = [vp, vp, sz, vp, sz, C.c_uint, i32, i32, C.c_uint, _f32p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.hnsw_gpu_index_abort.argtypes = [vp]
    L.hnsw_gpu_abort_all.restype = i32
    L.hnsw_gpu_ctx_create.argtypes = [vp, C.POINTER(vp)]
    L.hnsw_gpu_ctx_destroy.restype = None
    L.hnsw_gpu_ctx_destroy.argtypes = [vp]
    L.hnsw_gpu_search_batch_ctx.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hnsw_gpu_ctx_search_ms.argtypes = [vp, C.c_uint, _f32p]
    L.hnsw_gpu_search_batch_ctx_host.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.hnsw_gpu_search_batch_ctx_flags.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp, vp]
    L.hnsw_gpu_ctx_idle.argtypes = [vp]
    L.hnsw_gpu_host_alloc.restype = vp
    L.hnsw_gpu_host_alloc.argtypes = [sz]
    L.hnsw_gpu_host_free.restype = None
    L.hnsw_gpu_host_free.argtypes = [vp]
    L.hnsw_gpu_dist_batch.argtypes = [i32, vp, vp, sz, sz, vp]
    L.hnsw_gpu_dist_batch_dev.argtypes = [i32, vp, vp, sz, sz, sz, vp, vp]
    L.hnsw_gpu_bruteforce_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.hnsw_gpu_bruteforce_mfma_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.hnsw_gpu_last_bruteforce_gemm_ms.restype = C.c_float
    L.hnsw_gpu_last_bruteforce_clock_mhz.restype = C.c_double
    L.hnsw_gpu_last_bruteforce_tile.restype = C.c_int
    L.hnsw_gpu_device_wait.argtypes = [C.c_int, vp]
    L.hnsw_gpu_shared_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(vp), C.c_char_p]
    L.hnsw_gpu_shared_open.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp)]
    L.hnsw_gpu_shared_close.argtypes = [C.c_int, vp]
    L.hnsw_gpu_shared_free.argtypes = [C.c_int, vp]
    L.hnsw_gpu_merge_topk_dev.argtypes = [i32, vp, vp, sz, sz, sz, vp, vp, vp, vp]
    L.hnsw_gpu_merge_topk_strided_dev.argtypes = [i32, vp, sz, vp, sz, sz, sz, sz, vp, vp, vp, vp]
    L.hnsw_gpu_last_search_kernel.argtypes = [vp, C.c_char_p, sz]
    L.hnsw_gpu_team_counters.argtypes = [vp, _u32p]
    L.hnsw_gpu_last_search_clock_mhz.argtypes = [vp, C.POINTER(C.c_double)]
    L.hnsw_gpu_index_placement.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.hnsw_gpu_gather_roof.argtypes = [vp, i32, i32, C.c_uint, _f32p]
    L.hnsw_gpu_sharded_create.argtypes = [C.POINTER(vp), sz, C.POINTER(vp)]
    L.hnsw_gpu_sharded_destroy.restype = None
    L.hnsw_gpu_sharded_destroy.argtypes = [vp]
    L.hnsw_gpu_sharded_nshards.restype = sz
    L.hnsw_gpu_sharded_nshards.argtypes = [vp]
    L.hnsw_gpu_sharded_search_dev.argtypes = [vp, vp, sz, sz, vp, vp, vp, vp]
    L.hnsw_gpu_sharded_search.argtypes = [vp, vp, sz, sz, vp, vp, vp]
    L.hnsw_gpu_sharded_last_ms.argtypes = [vp, _f32p, _f32p, _f32p, C.POINTER(C.c_int)]
    L.hnsw_gpu_last_batch_ms.argtypes = [vp, _f32p]
    L.hnsw_gpu_ctx_set_walkers.argtypes = [vp, C.c_uint]
    L.hnsw_gpu_device_blocks.argtypes = [i32]
    L.hnsw_gpu_stream_open.argtypes = [vp, sz, sz, C.c_uint, C.POINTER(vp)]
    L.hnsw_gpu_stream_buffers.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.hnsw_gpu_stream_publish.argtypes = [vp, C.c_uint32]
    L.hnsw_gpu_stream_alive.argtypes = [vp]
    L.hnsw_gpu_stream_close.argtypes = [vp]
    L.hnsw_gpu_stream_abandon.argtypes = [vp]
    L.hnsw_gpu_config_set.argtypes = [C.c_char_p, C.c_char_p]
    L.hnsw_gpu_config_get.argtypes = [C.c_char_p, C.POINTER(C.c_longlong)]
    L.hnsw_gpu_config_reload.restype = None
    _gpu = _EnvSyncLib(L) if _env_sync_on() else L
    return _gpu


def shim_lib():
    """libembedding_gpu.so: the reference's four symbols.  Its host callbacks must already
    be resolvable in the global scope when a search through it is made."""
    global _shim
    if _shim is not None:
        return _shim
    gpu_lib()
    if not os.path.exists(_build.SHIM_LIB):
        raise LibraryMissing(f"{_build.SHIM_LIB} is missing: run __graft_entry__.build()")
    L = C.CDLL(_build.SHIM_LIB, mode=C.RTLD_GLOBAL | os.RTLD_LAZY)
    MP = C.c_void_p          # HnswMetadata*: any ctypes image of the struct may be passed
    L.hnsw_dist_func.restype = C.c_float
    L.hnsw_dist_func.argtypes = [C.c_int, _f32p, _f32p, C.c_size_t]
    L.hnsw_init_dist_func.restype = None
    L.hnsw_search.restype = C.c_bool
    L.hnsw_search.argtypes = [MP, _f32p, C.POINTER(C.c_size_t), C.POINTER(_u64p)]
    L.hnsw_bind_point.restype = C.c_bool
    L.hnsw_bind_point.argtypes = [MP, _f32p, C.c_uint32]
    L.hnsw_gpu_shim_snapshot.argtypes = [MP, C.POINTER(C.c_void_p)]
    L.hnsw_gpu_shim_attach.argtypes = [MP, C.c_void_p]
    L.hnsw_gpu_shim_detach.argtypes = [MP]
    _shim = L
    return L


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = gpu_lib().hnsw_gpu_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")


# ---------------------------------------------------------------------------------------------------------------------
# Knobs.  The library resolves its configuration ONCE (operational knobs from the environment at first use) and never calls
# getenv on a call path; a host that wants another value later says so through hnsw_gpu_config_set.  The test tiers and the
# experiment scripts flip kernel forms inside one process by changing os.environ: with PGEMB_ENV_SYNC=1 (tests/conftest.py and
# those scripts set it) the entry points of pg_embedding_amd forward every HNSW_GPU_* change to the library before they launch.
# ---------------------------------------------------------------------------------------------------------------------
def config_set(name: str, value) -> None:
    """Set (value = None: back to its default) one knob of the library — INTEGRATION.md lists them."""
    rc = gpu_lib().hnsw_gpu_config_set(name.encode(), None if value is None else str(value).encode())
    check(rc, f"hnsw_gpu_config_set({name})")


def config_get(name: str):
    """The knob's value, or None while it is at its default."""
    v = C.c_longlong(0)
    rc = gpu_lib().hnsw_gpu_config_get(name.encode(), C.byref(v))
    if rc == 1:
        return None
    check(rc, f"hnsw_gpu_config_get({name})")
    return v.value


_env_seen: dict = {}
_NOT_KNOBS = ("HNSW_GPU_WATCHDOG_S",)          # read once by the library itself, at its first workspace


def sync_env(force: bool = False) -> None:
    """Forward the HNSW_GPU_* variables that changed since the last call (set, changed or removed) to the library."""
    if not (_env_sync_on() or force):
        return
    global _env_seen
    now = {k: v for k, v in os.environ.items() if k.startswith("HNSW_GPU_") and k not in _NOT_KNOBS}
    if now == _env_seen:
        return
    L = gpu_lib()
    for k in set(now) | set(_env_seen):
        if now.get(k) != _env_seen.get(k):
            v = now.get(k)
            rc = L.hnsw_gpu_config_set(k.encode(), None if v is None else v.encode())
            if rc != 0 and v is not None:
                msg = L.hnsw_gpu_last_error()
                raise RuntimeError(f"{k}={v}: {msg.decode() if msg else rc} (knobs of rejected experiments exist only in -DHNSW_EXPERIMENT builds)")
    _env_seen = now
