"""oracle/lsap.c (restatement of scipy's rectangular LSAP, A2/models/matcher.py:6,246) vs scipy itself."""
import numpy as np
import pytest
from scipy.optimize import linear_sum_assignment as scipy_lsa

from oracle.lsap import linear_sum_assignment as oracle_lsa


@pytest.mark.parametrize("nr,nc", [(1, 1), (5, 5), (37, 300), (300, 37), (120, 300), (200, 576), (450, 300),
                                   (64, 64), (3, 900)])
@pytest.mark.parametrize("kind", ["float", "f32", "int_ties", "many_ties"])
def test_matches_scipy(nr, nc, kind):
    rng = np.random.default_rng(nr * 1000 + nc)
    for trial in range(3):
        if kind == "float":
            c = rng.standard_normal((nr, nc))
        elif kind == "f32":
            c = rng.standard_normal((nr, nc)).astype(np.float32)
        elif kind == "int_ties":
            c = rng.integers(0, 10, (nr, nc)).astype(np.float64)
        else:
            c = rng.integers(0, 2, (nr, nc)).astype(np.float64)
        a, b = oracle_lsa(c)
        sa, sb = scipy_lsa(c)
        assert np.array_equal(a, sa) and np.array_equal(b, sb)


def test_edge_cases():
    a, b = oracle_lsa(np.zeros((4, 4)))
    assert list(a) == [0, 1, 2, 3] and list(b) == [0, 1, 2, 3]           # all ties -> identity
    a, b = oracle_lsa(np.zeros((0, 5)))
    assert len(a) == 0 and len(b) == 0
    with pytest.raises(ValueError):
        oracle_lsa(np.array([[1.0, np.nan], [0.0, 1.0]]))
    with pytest.raises(ValueError):
        oracle_lsa(np.array([[1.0, -np.inf], [0.0, 1.0]]))
    c = np.array([[np.inf, 1.0], [np.inf, 2.0]])
    with pytest.raises(ValueError):
        oracle_lsa(c)
    with pytest.raises(ValueError):
        scipy_lsa(c)
