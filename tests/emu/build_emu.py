"""Build tests/_build/libhnsw_gpu_simt.so: the product's own sources (pg_embedding_amd/csrc/hnsw_gpu.hip, gpu_*.hip + device headers),
unmodified, compiled for the host against the SIMT emulator in tests/emu/hip/hip_runtime.h.  Test infrastructure only.

The one textual change: an inline-asm register pin with the AMDGPU constraint "+s" (scalar register), which no host compiler
knows, becomes "+r" (rounds 2-5 had one; round 6's kernels have none).  Everything else is the product's text.

build()                       the shipped sources
build_tree(csrc, tag, edit)   another source tree (e.g. with scripts/pending applied), or the shipped one with a textual
                              edit (a deliberately broken variant that shows a test has teeth)"""
import os
import shutil
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "pg_embedding_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "emu")
OUT = os.path.join(ROOT, "tests", "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
# the host-side translation units of the product library (pg_embedding_amd/build.py, HOST_UNITS)
HOST_HIP = ("hnsw_gpu.hip", "gpu_search.hip", "gpu_stream.hip", "gpu_scan.hip", "gpu_build.hip", "gpu_sharded.hip", "gpu_diag.hip")


def _deps(csrc):
    return [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".h", ".hip"))] + \
           [os.path.join(EMU, "hip", "hip_runtime.h"), os.path.join(EMU, "sort_pairs_emu.cpp"), os.path.abspath(__file__)]


def build_tree(csrc=CSRC, tag="", edit=None, force=False, verbose=False):
    lib = os.path.join(OUT, f"libhnsw_gpu_simt{('_' + tag) if tag else ''}.so")
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in _deps(csrc)):
        return lib
    src = os.path.join(OUT, "emu_src" + (("_" + tag) if tag else ""))
    shutil.rmtree(src, ignore_errors=True)
    os.makedirs(src)
    pins = 0
    for f in sorted(os.listdir(csrc)):                      # copies, so that quoted includes resolve among them
        if f.endswith(".h") or f in HOST_HIP or f == "search_inst.hip":
            txt = open(os.path.join(csrc, f)).read()
            pins += txt.count('"+s"')
            txt = txt.replace('"+s"', '"+r"')
            if edit:
                txt = edit(f, txt)
            open(os.path.join(src, f[:-4] + "_emu.cpp" if f.endswith(".hip") else f), "w").write(txt)
    cxx = CLANG if os.path.exists(CLANG) else "clang++"
    flags = ["-x", "c++", "-O0", "-std=c++17", "-mavx2", "-mfma", "-ffp-contract=off", "-fPIC", "-pthread",
             "-Wno-unused-value", "-Wno-pass-failed", "-Wno-unknown-attributes",
             "-I", src, "-I", EMU, "-I", os.path.join(ROOT, "include")]
    # the product's translation units (build.py): the host code and one unit per load shape of the search kernels, in parallel
    units = [(f[:-4] + "_emu", os.path.join(src, f[:-4] + "_emu.cpp"), []) for f in HOST_HIP if os.path.exists(os.path.join(src, f[:-4] + "_emu.cpp"))]
    units += [("sort_pairs_emu", os.path.join(EMU, "sort_pairs_emu.cpp"), [])]
    if os.path.exists(os.path.join(src, "search_inst_emu.cpp")):
        units += [(f"search_inst_{k}", os.path.join(src, "search_inst_emu.cpp"), [f"-DSEARCH_INST_SHAPE={k}"]) for k in range(1, 6)]
    procs = []
    for name, path, extra in units:
        cmd = [cxx] + flags + extra + ["-c", path, "-o", os.path.join(src, name + ".o")]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    for cmd, pr in procs:
        _, err = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError("SIMT emulator build failed:\n" + " ".join(cmd) + "\n" + err[-6000:])
    cmd = [cxx, "-shared", "-pthread"] + [os.path.join(src, name + ".o") for name, _, _ in units] + ["-o", lib]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("SIMT emulator link failed:\n" + r.stderr[-6000:])
    return lib


def build(force=False, verbose=False):
    return build_tree(force=force, verbose=verbose)


def build_with_patch(patch, tag, force=False):
    """the shipped csrc with `patch` (a git diff against the repository root) applied to a copy"""
    lib = os.path.join(OUT, f"libhnsw_gpu_simt_{tag}.so")
    if not force and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(s) for s in _deps(CSRC) + [patch]):
        return lib
    tree = os.path.join(OUT, "emu_tree_" + tag)
    shutil.rmtree(tree, ignore_errors=True)
    shutil.copytree(CSRC, os.path.join(tree, "pg_embedding_amd", "csrc"))
    subprocess.run(["git", "apply", "--include=pg_embedding_amd/csrc/*", os.path.abspath(patch)], cwd=tree, check=True)
    return build_tree(os.path.join(tree, "pg_embedding_amd", "csrc"), tag, force=True)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
