"""The control words of the beam kernel's team form (device_search.h: TeamCtl) under schedules the device suites cannot
force: tests/experiments/team_protocol_model.cpp, every wave a host thread, every LDS word an atomic.

What ships must be clean: a walking wave that clears its helper bits before it opens a walk never consumes a package that
was scored against its previous query.  The slice-job part models the ROUND-2 form of that protocol (one shared counter of
finished slices, an unbounded wait): with the clear and the job counter read before the helper's bit becomes visible no job
is left waiting, without either one a job is — the run that never returned at the end of round 2.  It is kept as the record
of WHY the shipped protocol (device_search.h, banner at TeamCtl: per-helper completion words carrying job number and walking
wave, every wait bounded with a correct fallback) keeps both orderings although it no longer depends on them for progress;
the shipped protocol itself runs under real wave schedules in tests/test_simt_emu.py and on the device in
tests/test_gpu_team_stress.py.  The variants WITHOUT the two orderings are run as well and reported (they do go wrong within
a few thousand walks on an idle machine — which is how the orderings were found to be necessary — but a test must not depend
on a race being lost in time)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "experiments", "team_protocol_model.cpp")
OUT = os.path.join(ROOT, "tests", "_build", "team_protocol_model")


@pytest.fixture(scope="module")
def model():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", SRC, "-o", OUT], check=True)
    return OUT


def run(model, clear, early, slices, walks, seed, limit_ms=20000):
    r = subprocess.run([model, str(clear), str(early), str(slices), str(walks), str(seed), str(limit_ms)], capture_output=True, text=True, timeout=300)
    fields = dict(kv.split("=") for kv in r.stdout.split())
    return r.returncode, {k: int(v) for k, v in fields.items()}


@pytest.mark.parametrize("seed", [3, 17])
def test_shipped_team_protocol_never_reads_a_package_of_the_previous_query(model, seed):
    rc, f = run(model, 1, 1, 0, 1500, seed)
    assert rc == 0 and f["stale"] == 0 and f["hangs"] == 0
    assert f["pk"] > 1000                                   # packages were consumed at all


@pytest.mark.parametrize("seed", [5, 23])
def test_round2_slice_jobs_complete_with_both_orderings(model, seed):
    rc, f = run(model, 1, 1, 1, 1500, seed)
    assert rc == 0 and f["stale"] == 0 and f["hangs"] == 0
    assert f["jobs"] > 500


def test_variants_without_the_two_orderings_are_reported(model, capsys):
    lines = []
    for clear, early, slices in ((0, 1, 0), (1, 0, 1), (0, 1, 1)):
        rc, f = run(model, clear, early, slices, 2000, 9, limit_ms=1500)
        lines.append(f"clear={clear} early={early} slices={slices}: stale={f['stale']} hangs={f['hangs']} jobs={f['jobs']} pk={f['pk']}")
    with capsys.disabled():
        print("\n[team protocol model] " + "; ".join(lines))
