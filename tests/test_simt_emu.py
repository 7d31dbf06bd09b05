"""The kernels' OWN source on the CPU: pg_embedding_amd/csrc/hnsw_gpu.hip and its device headers, unmodified, compiled for the
host against a SIMT emulator (tests/emu/hip/hip_runtime.h: a wavefront = an OS thread, its 64 lanes = coroutines that meet at
every cross-lane operation) and compared with the oracle bit for bit.

Test infrastructure: the product is the hipcc build for gfx950 and has no CPU path (tests/test_abi.py checks that); nothing
outside tests/ can reach the emulated library.  What it adds to the CPU tier, which otherwise only sees the oracle and the host
logic: (1) every kernel form the host can pick — beam / two-set register / LDS form, one-wave and team form, the five row
shapes, three metrics — walks and emits exactly as the oracle does; (2) the lock-free protocols between the waves of a block
run under real preemptive schedules, including the one no device suite produces: a wave that walks many queries while its
siblings help; (3) the same for the change that waits in scripts/pending for a device."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import build_emu                                           # noqa: E402

RUN = os.path.join(ROOT, "tests", "emu", "run_emu_case.py")
PENDING = os.path.join(ROOT, "scripts", "pending", "slice_helpers_and_bulk_append.patch")


def run_case(case, lib, env=None, timeout=900):
    r = subprocess.run([sys.executable, RUN, case, lib], capture_output=True, text=True, timeout=timeout, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


@pytest.fixture(scope="module")
def emu_lib():
    return build_emu.build()


def test_every_kernel_form_walks_and_emits_like_the_oracle(emu_lib):
    res = run_case("forms", emu_lib)
    bad = [r for r in res if r["wrong"]]
    assert not bad, bad
    kernels = {r["kernel"] for r in res}
    for needle in ("hnsw_search_kernel_beam<0, pgemb::Shape2x2, 2, false>", "hnsw_search_kernel_reg<", "hnsw_search_kernel_lds<",
                   "Shape12x2, 2, true>", "Shape4x2", "kernel_beam<1,", "kernel_beam<2,"):
        assert any(needle in k for k in kernels), (needle, sorted(kernels))


def test_the_other_kernels_behind_the_c_abi(emu_lib):
    """serial device insert == the oracle's graph bytes, the walk's pop sequence, vacuum flags, a batched build that the
    search finds its way in; and the distance entry points: the device tier's own test file, unchanged, on the emulated library"""
    res = run_case("others", emu_lib)
    recall = res.pop("batched_insert_recall_at_10")
    assert all(v == 0 for v in res.values()), res
    assert recall > 0.6, recall
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_dist.py"), "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=900, env=dict(os.environ, PGEMB_GPU_LIB=emu_lib))
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-1500:], r.stderr[-500:])


def test_a_wave_that_walks_many_queries_with_helpers_attached(emu_lib):
    """one wave, 32 neighbouring queries, seven helpers (with and without schedule jitter; two waves with three helpers each)"""
    res = run_case("second_walk", emu_lib)
    assert all(r["wrong"] == 0 and "true>" in r["kernel"] for r in res), res


def test_the_same_without_the_helper_bit_clear_is_reported(capsys):
    """The schedule above is what the clear at the start of a walk (device_search.h) is for: without that one line a good part
    of the answers is wrong.  Reported, not asserted — it is a race, and a test must not depend on losing one."""
    def drop_clear(name, txt):
        if name == "device_search.h":
            assert txt.count("if (TEAM) ctl[wib].helpers = 0u;") == 1
            txt = txt.replace("if (TEAM) ctl[wib].helpers = 0u;", ";")
        return txt
    lib = build_emu.build_tree(tag="noclear", edit=drop_clear)
    res = run_case("second_walk", lib)
    with capsys.disabled():
        print("\n[simt emulator] without the helper-bit clear: " + "; ".join(f"{r['wrong']} of {r['walks']} answers wrong" for r in res))


@pytest.mark.parametrize("spec", ["5", "0"])
def test_pending_slice_helpers_and_hop_wide_append_are_exact_under_emulation(spec):
    """scripts/pending (not shipped: it waits for a device run): helpers scoring slices of the walking wave's rows, one-step
    append below ef — every kernel form still equals the oracle, and the many-walks schedule completes (no job left waiting)
    with five of seven helpers speculating (5: the measured setting) and none (0: all of them take slices)."""
    lib = build_emu.build_with_patch(PENDING, "pending")
    env = {"HNSW_GPU_TEAM_SPEC": spec}
    res = run_case("second_walk", lib, env, timeout=600)
    assert all(r["wrong"] == 0 for r in res), res
    if spec == "5":
        res = run_case("forms", lib, dict(env, EMU_FORMS_QUICK="1"))
        assert not [r for r in res if r["wrong"]], [r for r in res if r["wrong"]]
