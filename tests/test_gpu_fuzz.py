"""A slice of the randomised parity sweep (tests/experiments/fuzz_parity.py): random dims / m / ef /
metric / vacuum flags / ties, search and serial insert bit-exact against the oracle."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _fuzz():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "fuzz_parity.py")
    spec = importlib.util.spec_from_file_location("fuzz_parity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_random_configurations(seed):
    fz = _fuzz()
    rng = np.random.default_rng(seed)
    for i in range(12):
        assert fz.one_case(rng, 7000000 + 1000 * seed + i)


def _fuzz_mfma():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "experiments", "fuzz_mfma.py")
    spec = importlib.util.spec_from_file_location("fuzz_mfma", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("seed", [21, 22])
def test_random_exhaustive_scorer_configurations(seed):
    """A slice of tests/experiments/fuzz_mfma.py: the MFMA filter + canonical re-score == the canonical scan (ids, distance bits) over
    random table sizes around tile edges, rows that end inside a K step, one to several query tiles, ties, duplicates, zero distances."""
    fz = _fuzz_mfma()
    rng = np.random.default_rng(seed)
    for i in range(8):
        assert fz.one_case(rng, 9000000 + 1000 * seed + i)
