"""Oracle restatements of third-party routines against the installed SciPy."""
import numpy as np
from scipy.optimize import linear_sum_assignment
from scipy.ndimage import gaussian_filter1d

from oracle import cpu_ref as O


def _same(cost):
    r0, c0 = linear_sum_assignment(cost)
    r1, c1 = O.lsap(cost)
    assert np.array_equal(r0, r1) and np.array_equal(c0, c1), (cost, r0, c0, r1, c1)


def test_lsap_known_answers():
    _same(np.zeros((3, 3))); _same(np.zeros((2, 4))); _same(np.zeros((4, 2)))
    _same(np.array([[1, 1, 2], [1, 1, 2], [2, 2, 1.0]]))
    _same(-np.array([[.9, 0, 0], [0, 0, 0], [0, .8, 0]]))
    r, c = O.lsap(np.zeros((0, 3)))
    assert len(r) == 0 and len(c) == 0


def test_lsap_fuzz():
    rng = np.random.default_rng(0)
    for it in range(600):
        n, m = rng.integers(1, 9, size=2)
        kind = it % 4
        if kind == 0:
            cost = rng.normal(size=(n, m))
        elif kind == 1:                      # tie-heavy small integers
            cost = rng.integers(0, 3, size=(n, m)).astype(float)
        elif kind == 2:                      # association-like: mostly exact zeros, a few negative entries
            cost = np.zeros((n, m))
            for _ in range(min(n, m)):
                cost[rng.integers(n), rng.integers(m)] = -rng.uniform(0.1, 1)
        else:                                # zero rows/cols mixed with reals
            cost = rng.uniform(0, 1, size=(n, m)) * (rng.uniform(size=(n, 1)) > 0.4)
        _same(cost)


def test_gaussian_last_sample():
    rng = np.random.default_rng(1)
    for L in range(0, 13):
        hist = rng.normal(size=(L, 17, 3))
        raw = rng.normal(size=(17, 3))
        for sigma, arm in ((0.3, 0.8), (0.6, 0.8), (1.3, 2.0)):
            seq = np.concatenate([hist, raw[None]], axis=0)
            exp = raw.copy()
            na = [0, 1, 2, 3, 4, 5, 6, 7, 8, 11, 12, 13, 14, 15, 16]
            exp[na] = gaussian_filter1d(seq[:, na, :].T, sigma=sigma, mode='reflect')[:, :, -1].T
            exp[[9, 10]] = gaussian_filter1d(seq[:, [9, 10], :].T, sigma=arm, mode='reflect')[:, :, -1].T
            got = O.smooth_last(hist, raw, sigma, arm)
            np.testing.assert_allclose(got, exp, rtol=1e-13, atol=1e-13)
