"""Device distance kernels vs the oracle (bit-exact) and vs the reference binary (1e-5 rel)."""
import numpy as np
import pytest

import oracle
import pg_embedding_amd as pg
from util import REL_TOL, bits, rel_err

pytestmark = pytest.mark.gpu

FUNCS = [pg.DIST_L2, pg.DIST_COSINE, pg.DIST_MANHATTAN]
DIMS = [1, 3, 4, 5, 63, 64, 65, 100, 128, 255, 256, 257, 768, 1000, 1536, 2000]


@pytest.mark.parametrize("func", FUNCS)
@pytest.mark.parametrize("dim", DIMS)
def test_dist_batch_bit_exact_vs_oracle(func, dim):
    rng = np.random.default_rng(dim * 7 + func)
    q = rng.standard_normal(dim).astype(np.float32)
    rows = rng.standard_normal((257, dim)).astype(np.float32)
    rows[0] = q                      # zero distance
    rows[1] = -q                     # cosine distance 2
    rows[2] = q * np.float32(1e-3)   # same direction, tiny norm
    rows[3] *= np.float32(1e4)       # large magnitudes
    got = pg.dist_batch(func, q, rows)
    want = oracle.port_dist_many(func, q, rows)
    assert (bits(got) == bits(want)).all(), f"max rel {rel_err(got, want).max()}"


@pytest.mark.parametrize("func", FUNCS)
@pytest.mark.parametrize("dim", [3, 128, 768, 1536])
def test_dist_batch_within_tolerance_of_reference(func, dim):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(dim + 31 * func)
    q = rng.standard_normal(dim).astype(np.float32)
    rows = (rng.standard_normal((500, dim)) + 0.2).astype(np.float32)
    got = pg.dist_batch(func, q, rows)
    ref = oracle.ref_dist_many(func, q, rows)
    # tolerance stated by the north star: 1e-5 relative (with a 1e-6 absolute floor)
    assert rel_err(got, ref).max() <= REL_TOL


def test_toy_known_answers():
    """README / knn.sql rows vs {3,3,3}: values quoted in BASELINE.md §4."""
    rows = np.array([[1, 2, 3], [1, 2, 4], [1, 1, 1], [0, 1, 2]], np.float32)
    q = np.array([3, 3, 3], np.float32)
    l2 = pg.dist_batch(pg.DIST_L2, q, rows)
    cos = pg.dist_batch(pg.DIST_COSINE, q, rows)
    man = pg.dist_batch(pg.DIST_MANHATTAN, q, rows)
    np.testing.assert_allclose(l2, [2.236068, 2.44949, 3.464102, 3.741657], rtol=1e-6)
    np.testing.assert_allclose(cos, [0.0741799, 0.1180829, 0.0, 0.2254033], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(man, [3, 4, 6, 6])


def test_sql_scalar_functions():
    a = np.array([1, 2, 3], np.float32)
    b = np.array([3, 3, 3], np.float32)
    assert abs(pg.l2_distance(a, b) - 2.236068) < 1e-6
    assert abs(pg.cosine_distance(a, b) - 0.0741799) < 1e-6
    assert pg.manhattan_distance(a, b) == 3.0
    with pytest.raises(ValueError):
        pg.l2_distance(a, b[:2])    # "Different array dimensions", embedding.c:1030-1035


def test_integer_data_is_exact_against_reference():
    """SIFT-like integer coordinates: every order of summation gives the same L2."""
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    from pg_embedding_amd.datasets import sift_like
    X = sift_like(300, 128, seed=3)
    got = pg.dist_batch(pg.DIST_L2, X[0], X)
    ref = oracle.ref_dist_many(pg.DIST_L2, X[0], X)
    assert (bits(got) == bits(ref)).all()
