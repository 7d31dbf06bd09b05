"""Compile the gfx950 libraries in-tree (pg_embedding_amd/lib/*.so).

    libhnsw_gpu.so       HIP kernels + additive C API   (include/hnsw_gpu.h)
    libembedding_gpu.so  the reference's four symbols   (include/hnsw_abi.h) on top of it
    libembedding_gpuc.so the same four symbols as a client of hnsw_gpu_server (no HIP linked)
    bin/hnsw_gpu_server  the GPU-owning batching server (include/hnsw_gpu_server.h)

hipcc cross-compiles for gfx950 without a GPU.  The .so files are git-ignored but travel
to the GPU box with the tree.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
INC = os.path.join(ROOT, "include")

GPU_LIB = os.path.join(LIBDIR, "libhnsw_gpu.so")
SHIM_LIB = os.path.join(LIBDIR, "libembedding_gpu.so")
CLIENT_LIB = os.path.join(LIBDIR, "libembedding_gpuc.so")
BINDIR = os.path.join(PKG, "bin")
SERVER_BIN = os.path.join(BINDIR, "hnsw_gpu_server")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    # the canonical summation order (oracle/hnsw_port.c) relies on explicit FMAs only
    "-ffp-contract=off",
]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm with gfx950 support)")


def _digest(sources, cmd) -> str:
    h = hashlib.sha256(" ".join(cmd).encode())
    for s in sorted(sources):
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp(target: str) -> str:
    return target + ".stamp"


def _current(target: str, digest: str) -> bool:
    """Up to date = the artefact exists and was built from exactly these source BYTES with this command (a stamp file
    beside it holds the digest taken when its build STARTED).  Not by modification time: a source edited while its build
    was running would look older than the artefact, and a tree copied to another box (gpurun) gets new times."""
    try:
        with open(_stamp(target)) as f:
            return os.path.exists(target) and f.read().strip() == digest
    except OSError:
        return False


def _built(target: str, digest: str) -> None:
    with open(_stamp(target), "w") as f:
        f.write(digest + "\n")


def _run(cmd) -> None:
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)



# the host-side units of libhnsw_gpu.so (csrc/gpu_host.h says what each holds)
HOST_UNITS = ("hnsw_gpu", "gpu_search", "gpu_stream", "gpu_scan", "gpu_build", "gpu_sharded", "gpu_diag")


# libhnsw_gpu.so = these translation units, compiled in parallel and linked: the host code + its small kernels, the pair sort,
# and the search kernels of one load shape each (csrc/search_inst.hip, -DSEARCH_INST_SHAPE=n) — as one unit the 220 search
# kernel instantiations took hipcc five and a half minutes, like this the library builds in about two.
def _gpu_units(defines):
    units = [(u, u + ".hip", []) for u in HOST_UNITS] + [("sort_pairs", "sort_pairs.hip", [])]
    shapes = range(1, 7) if "HNSW_EXPERIMENT" in defines else range(1, 6)
    return units + [(f"search_inst_{k}", "search_inst.hip", [f"-DSEARCH_INST_SHAPE={k}"]) for k in shapes]


def _build_gpu_lib(target, sources, defines, force=False, verbose=False):
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + [f"-D{d}" for d in defines]
    d = _digest(sources, flags + [os.path.basename(target)])
    if not force and _current(target, d):
        return
    objdir = os.path.join(LIBDIR, "obj", os.path.basename(target))
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for name, src, extra in _gpu_units(defines):
        obj = os.path.join(objdir, name + ".o")
        cmd = [_hipcc()] + flags + extra + ["-I", INC, "-I", CSRC, "-c", os.path.join(CSRC, src), "-o", obj]
        ud = _digest(sources, flags + extra + [src])         # per-object stamp: an unchanged unit is not recompiled
        if not force and _current(obj, ud):
            procs.append((obj, None, None, ud))
            continue
        if verbose:
            print(" ".join(cmd))
        procs.append((obj, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True), ud))
    for obj, cmd, pr, ud in procs:
        if pr is None:
            continue
        out, err = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("build failed: " + " ".join(cmd) + "\n" + out + err)
        _built(obj, ud)
    _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + [o for o, _, _, _ in procs] + ["-o", target])
    _built(target, d)

def build(force: bool = False, verbose: bool = False) -> None:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(BINDIR, exist_ok=True)
    hdrs = [os.path.join(INC, h) for h in ("hnsw_abi.h", "hnsw_gpu.h", "hnsw_gpu_diag.h", "hnsw_gpu_shim.h", "hnsw_gpu_server.h")]
    host_only = ("hgs_io.h", "host_walk.h", "host_dist.h", "shim_cache.h")
    gpu_src = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and f not in host_only] + hdrs

    def step(target, sources, cmd):
        d = _digest(sources, [os.path.relpath(c, ROOT) if c.startswith(ROOT + os.sep) else c for c in cmd])   # (the tree may live anywhere)
        if not force and _current(target, d):
            return
        if verbose:
            print(" ".join(cmd))
        _run(cmd)
        _built(target, d)

    _build_gpu_lib(GPU_LIB, gpu_src, [], force, verbose)
    host_dist = os.path.join(CSRC, "host_dist.h")
    gpu_stamp = [_stamp(GPU_LIB)]                            # the host libraries link against the device library
    shim_src = [os.path.join(CSRC, "embedding_shim.cpp"), os.path.join(CSRC, "host_walk.h"), os.path.join(CSRC, "shim_cache.h"), host_dist] + hdrs
    step(SHIM_LIB, shim_src + gpu_stamp,
         ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-I", INC, "-I", CSRC,
          os.path.join(CSRC, "embedding_shim.cpp"), "-o", SHIM_LIB, "-L", LIBDIR, "-lhnsw_gpu", "-Wl,-rpath,$ORIGIN", "-lpthread"])
    io_h = os.path.join(CSRC, "hgs_io.h")
    client_src = [os.path.join(CSRC, "remote_client.cpp"), io_h, os.path.join(CSRC, "host_walk.h"), host_dist] + hdrs
    step(CLIENT_LIB, client_src,
         ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-ffp-contract=off", "-I", INC, "-I", CSRC,
          os.path.join(CSRC, "remote_client.cpp"), "-o", CLIENT_LIB, "-lpthread"])
    server_src = [os.path.join(CSRC, "server_main.cpp"), io_h] + hdrs
    step(SERVER_BIN, server_src + gpu_stamp,
         ["g++", "-O2", "-std=c++17", "-Wall", "-I", INC, "-I", CSRC, os.path.join(CSRC, "server_main.cpp"), "-o", SERVER_BIN,
          "-L", LIBDIR, "-lhnsw_gpu", "-Wl,-rpath,$ORIGIN/../lib", "-lpthread"])


def build_variant(tag: str, defines) -> str:
    """An experiment build of libhnsw_gpu.so with extra -D flags (lib/variants/libhnsw_gpu_<tag>.so);
    PGEMB_GPU_LIB=<path> makes pg_embedding_amd load it instead of the product library."""
    vdir = os.path.join(LIBDIR, "variants")
    os.makedirs(vdir, exist_ok=True)
    out = os.path.join(vdir, f"libhnsw_gpu_{tag}.so")
    defines = list(defines)
    if any(d.split("=")[0] in ("HNSW_TEAM_COUNTERS", "HNSW_HOP_STAMPS") for d in defines) and "HNSW_EXPERIMENT" not in defines:
        defines.append("HNSW_EXPERIMENT")                   # diagnostic builds read every knob from the environment
    hdrs = [os.path.join(INC, h) for h in ("hnsw_abi.h", "hnsw_gpu.h", "hnsw_gpu_diag.h", "hnsw_gpu_shim.h", "hnsw_gpu_server.h")]
    host_only = ("hgs_io.h", "host_walk.h", "host_dist.h", "shim_cache.h")
    gpu_src = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")) and f not in host_only] + hdrs
    _build_gpu_lib(out, gpu_src, defines)
    import shutil
    shutil.rmtree(os.path.join(LIBDIR, "obj", os.path.basename(out)), ignore_errors=True)   # (9 MB of objects per variant, and they travel to the GPU box)
    return out


if __name__ == "__main__":
    import sys
    if len(sys.argv) > 2 and sys.argv[1] == "variant":      # build.py variant <tag> [DEFINE ...]
        print("built", build_variant(sys.argv[2], sys.argv[3:]))
        sys.exit(0)
    build(force=True, verbose=True)
    print("built", GPU_LIB, SHIM_LIB, CLIENT_LIB, SERVER_BIN)
