"""hnsw_gpu_server from Python: start/stop the GPU-owning server process and talk to it through
the client library (libembedding_gpuc.so, include/hnsw_gpu_server.h).

The server is the deployment shape for Postgres (one process per connection, one query per
hnsw_search call, embedding.c:284-343): backends link the client library, which has no HIP in it;
the server owns the device and coalesces the backends' concurrent searches into batch launches.
Nothing here computes: ctypes plumbing only.
"""
from __future__ import annotations

import ctypes as C
import os
import signal
import subprocess
import tempfile
import time
from typing import Optional, Tuple

import numpy as np

from . import build as _build
from ._lib import HnswMetadata, LibraryMissing

_u64p = C.POINTER(C.c_uint64)
_f32p = C.POINTER(C.c_float)


class hgs_stats(C.Structure):
    """include/hnsw_gpu_server.h: hgs_stats."""
    _fields_ = [(n, C.c_uint64) for n in (
        "connections", "connections_now", "searches", "batches", "max_batch", "search_errors",
        "uploads", "upload_bytes", "updates", "binds", "evictions", "mirrors", "mirror_elements",
        "batch_ns", "kernel_ns", "uptime_ns", "queue_ns", "walk_ns", "answer_ns", "shm_searches")]


HGS_ERR_NOKEY, HGS_ERR_STALE, HGS_ERR_IO = -21, -22, -23

_client = None


def client_lib(path: Optional[str] = None):
    """libembedding_gpuc.so with argtypes set.  Its drop-in symbols import the host's storage
    callbacks; the hnsw_gpu_remote_* calls bound here need none."""
    global _client
    if _client is not None and path is None:
        return _client
    p = path or _build.CLIENT_LIB
    if not os.path.exists(p):
        _build.build()
    if not os.path.exists(p):
        raise LibraryMissing(f"{p} is missing: run __graft_entry__.build()")
    L = C.CDLL(p, mode=os.RTLD_LAZY | os.RTLD_LOCAL)
    sz, u64, i32, vp = C.c_size_t, C.c_uint64, C.c_int, C.c_void_p
    MP = C.POINTER(HnswMetadata)
    L.hnsw_gpu_remote_last_error.restype = C.c_char_p
    L.hnsw_gpu_remote_connect.argtypes = [C.c_char_p]
    L.hnsw_gpu_remote_disconnect.restype = None
    L.hnsw_gpu_remote_lookup.argtypes = [u64, _u64p, C.POINTER(sz), C.POINTER(i32)]
    L.hnsw_gpu_remote_upload.argtypes = [MP, u64, u64, vp, sz]
    L.hnsw_gpu_remote_update.argtypes = [u64, u64, u64, MP, vp, sz, sz]
    L.hnsw_gpu_remote_search.argtypes = [u64, u64, _f32p, sz, sz, _u64p, _f32p, C.POINTER(sz)]
    L.hnsw_gpu_remote_link.argtypes = [u64, sz, sz, sz]
    L.hnsw_gpu_remote_export.argtypes = [u64, vp, sz]
    L.hnsw_gpu_remote_set_deleted.argtypes = [u64, C.c_uint32, i32]
    L.hnsw_gpu_remote_drop.argtypes = [u64]
    L.hnsw_gpu_remote_stats.argtypes = [C.POINTER(hgs_stats)]
    if path is None:
        _client = L
    return L


class RemoteError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"hnsw_gpu_server request failed ({code}): {msg}")
        self.code = code


class RemoteClient:
    """One connection (per calling thread) to a running hnsw_gpu_server."""

    def __init__(self, socket_path: str):
        self.L = client_lib()
        self._check(self.L.hnsw_gpu_remote_connect(socket_path.encode()))

    def _check(self, rc: int) -> None:
        if rc != 0:
            raise RemoteError(rc, (self.L.hnsw_gpu_remote_last_error() or b"").decode())

    def lookup(self, key: int) -> Tuple[bool, int, int]:
        gen, cnt, present = C.c_uint64(0), C.c_size_t(0), C.c_int(0)
        self._check(self.L.hnsw_gpu_remote_lookup(key, C.byref(gen), C.byref(cnt), C.byref(present)))
        return bool(present.value), int(gen.value), int(cnt.value)

    def upload(self, meta: HnswMetadata, key: int, gen: int, raw: np.ndarray, n: int) -> None:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.size == n * meta.size_data_per_element
        self._check(self.L.hnsw_gpu_remote_upload(C.byref(meta), key, gen, raw.ctypes.data if n else None, n))

    def update(self, meta: HnswMetadata, key: int, expect_gen: int, new_gen: int, raw: np.ndarray,
               first: int, count: int) -> None:
        raw = np.ascontiguousarray(raw, dtype=np.uint8)
        assert raw.size == count * meta.size_data_per_element
        self._check(self.L.hnsw_gpu_remote_update(key, expect_gen, new_gen, C.byref(meta), raw.ctypes.data, first, count))

    def search(self, key: int, q, ef: int, gen: int = 0):
        """One hnsw_search: (labels, distances), ascending by (distance, label)."""
        q = np.ascontiguousarray(q, dtype=np.float32)
        lab = np.empty(ef, np.uint64)
        dst = np.empty(ef, np.float32)
        n = C.c_size_t(0)
        self._check(self.L.hnsw_gpu_remote_search(key, gen, q.ctypes.data_as(_f32p), q.size, ef,
                                                  lab.ctypes.data_as(_u64p), dst.ctypes.data_as(_f32p), C.byref(n)))
        return lab[:n.value].copy(), dst[:n.value].copy()

    def link(self, key: int, first: int, count: int, max_batch: int = 0) -> None:
        self._check(self.L.hnsw_gpu_remote_link(key, first, count, max_batch))

    def export(self, key: int, nbytes: int) -> np.ndarray:
        out = np.empty(nbytes, np.uint8)
        self._check(self.L.hnsw_gpu_remote_export(key, out.ctypes.data, nbytes))
        return out

    def set_deleted(self, key: int, idx: int, deleted: bool = True) -> None:
        self._check(self.L.hnsw_gpu_remote_set_deleted(key, idx, int(deleted)))

    def drop(self, key: int) -> None:
        self._check(self.L.hnsw_gpu_remote_drop(key))

    def stats(self) -> dict:
        s = hgs_stats()
        self._check(self.L.hnsw_gpu_remote_stats(C.byref(s)))
        return {n: int(getattr(s, n)) for n, _ in hgs_stats._fields_}

    def close(self) -> None:
        self.L.hnsw_gpu_remote_disconnect()


class ServerProcess:
    """`with ServerProcess() as srv:` runs pg_embedding_amd/bin/hnsw_gpu_server on a private socket
    and stops it (SIGTERM) on exit.  `binary` lets the CPU tests substitute their test double."""

    def __init__(self, socket_path: Optional[str] = None, device: int = 0, dispatchers: int = 2,
                 max_batch: int = 16384, linger_us: int = 0, min_batch: int = 1, readers: int = 4, lanes: int = 3, binary: Optional[str] = None,
                 env: Optional[dict] = None, verbose: bool = False, start_timeout: float = 120.0, walkers: Optional[str] = None,
                 stream: bool = False, ring: int = 4096, shm_pollers: Optional[int] = None, shard_peers=None):
        self.binary = binary or _build.SERVER_BIN
        if not os.path.exists(self.binary):
            _build.build()
        if not os.path.exists(self.binary):
            raise LibraryMissing(f"{self.binary} is missing: run __graft_entry__.build()")
        self._tmp = None
        if socket_path is None:
            self._tmp = tempfile.mkdtemp(prefix="hgs_")
            socket_path = os.path.join(self._tmp, "s")
        self.socket_path = socket_path
        self.args = [self.binary, "--socket", socket_path, "--device", str(device), "--dispatchers", str(dispatchers),
                     "--max-batch", str(max_batch), "--readers", str(readers), "--lanes", str(lanes)]
        if walkers is not None:
            self.args += ["--walkers", str(walkers)]
        if stream:
            self.args += ["--stream", "1", "--ring", str(ring)]
        if shm_pollers is not None:
            self.args += ["--shm-pollers", str(shm_pollers)]
        if shard_peers:                          # this server is the FRONT of a row-sharded index: the peers hold the other shards (needs lanes=0)
            self.args += ["--shard-peers", ",".join(shard_peers)]
        if linger_us:
            self.args += ["--linger-us", str(linger_us), "--min-batch", str(min_batch)]
        if verbose:
            self.args += ["--verbose"]
        self.env = dict(os.environ, **(env or {}))
        self.start_timeout = start_timeout
        self.proc = None

    def start(self) -> "ServerProcess":
        r, w = os.pipe()
        self.proc = subprocess.Popen(self.args + ["--ready-fd", str(w)], pass_fds=(w,), env=self.env)
        os.close(w)
        deadline = time.time() + self.start_timeout
        got = b""
        os.set_blocking(r, False)
        try:
            while time.time() < deadline and b"READY" not in got:
                if self.proc.poll() is not None:
                    raise RuntimeError(f"hnsw_gpu_server exited with status {self.proc.returncode} "
                                       "(3 = no gfx950 device: there is no CPU path)")
                try:
                    chunk = os.read(r, 64)
                    if chunk:
                        got += chunk
                        continue
                except BlockingIOError:
                    pass
                time.sleep(0.01)
        finally:
            os.close(r)
        if b"READY" not in got:
            self.stop()
            raise RuntimeError("hnsw_gpu_server did not come up")
        return self

    def stop(self) -> int:
        rc = None
        if self.proc is not None:
            if self.proc.poll() is None:
                self.proc.send_signal(signal.SIGTERM)
                try:
                    self.proc.wait(timeout=30)
                except subprocess.TimeoutExpired:
                    self.proc.kill()
                    self.proc.wait()
            rc = self.proc.returncode
            self.proc = None
        if self._tmp:
            try:
                if os.path.exists(self.socket_path):
                    os.unlink(self.socket_path)
                os.rmdir(self._tmp)
            except OSError:
                pass
            self._tmp = None
        return rc

    def __enter__(self) -> "ServerProcess":
        return self.start()

    def __exit__(self, *exc) -> None:
        self.stop()
