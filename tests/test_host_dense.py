"""CPU: host-size dense kernels of the product (diaglib_amd/csrc/smalldense.cpp) against numpy/LAPACK.
These replace dsyev / dpotrf / dtrtri / norm_est of the reference (diaglib.f90:1708, 3261, 3310, 3447)."""
import numpy as np
import pytest

from diaglib_amd import capi

EPS = np.finfo(np.float64).eps


@pytest.mark.parametrize("n", [1, 2, 3, 13, 26, 39, 91, 160])
@pytest.mark.parametrize("uplo", ["u", "l"])
def test_syev_full(rng, n, uplo):
    a = rng.standard_normal((n, n)); s = a + a.T
    junk = rng.standard_normal((n, n))
    arg = np.triu(s) + np.tril(junk, -1) if uplo == "u" else np.tril(s) + np.triu(junk, 1)   # only one triangle is read
    w, v = capi.syev(arg, uplo)
    scale = np.abs(s).sum(1).max()
    assert np.abs(w - np.linalg.eigvalsh(s)).max() <= 50 * n * EPS * scale
    assert np.abs(s @ v - v * w).max() <= 50 * n * EPS * scale
    assert np.abs(v.T @ v - np.eye(n)).max() <= 50 * n * EPS
    assert np.all(np.diff(w) >= 0)


def _check_lowest(s, m, tol_scale=200):
    n = s.shape[0]
    w, v = capi.syev_lowest(np.triu(s), m, "u")
    scale = max(np.abs(s).sum(1).max(), 1e-300)
    assert np.abs(w[:m] - np.linalg.eigvalsh(s)[:m]).max() <= tol_scale * n * EPS * scale
    assert np.abs(s @ v - v * w[:m]).max() <= tol_scale * n * EPS * scale
    assert np.abs(v.T @ v - np.eye(m)).max() <= 200 * n * EPS


@pytest.mark.parametrize("n,m", [(40, 13), (52, 13), (91, 13), (106, 13), (260, 13), (420, 21), (111, 37), (33, 4)])
def test_syev_lowest_random(rng, n, m):
    a = rng.standard_normal((n, n))
    _check_lowest(a + a.T, m)


def test_syev_lowest_clusters_and_degeneracy(rng):
    n = 120
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([[1, 1, 1, 1 + 1e-13, 1 + 1e-9, 2, 2, 2 + 1e-6, 3, 3.0000001], np.linspace(5, 50, n - 10)])
    _check_lowest((q * lam) @ q.T, 13)
    _check_lowest(np.eye(80) * 2.0, 13)
    _check_lowest(np.diag(np.arange(100.0)), 21)
    s = np.zeros((150, 150)); b = rng.standard_normal((50, 50)); s[:50, :50] = b + b.T
    s[50:, 50:] = np.diag(rng.random(100) * 3)
    _check_lowest(s, 37)


def test_syev_lowest_graded_and_scaled(rng):
    n = 200
    d = np.diag(np.sort(rng.random(n)) * 1e4); b = rng.standard_normal((n, n)) * 0.1
    _check_lowest(d + b + b.T, 13)
    a = rng.standard_normal((100, 100))
    _check_lowest((a + a.T) * 1e-12, 13)
    _check_lowest((a + a.T) * 1e12, 13)


def test_syev_lowest_davidson_shaped(rng):
    """Arrow-like projected matrix: converged Ritz values on the diagonal + a dense border block."""
    n, k = 117, 13
    s = np.diag(np.concatenate([np.arange(2.0, 2.0 + n - k), np.zeros(k)]))
    bnd = rng.standard_normal((n, k)) * 1e-3
    s[:, n - k:] += bnd; s[n - k:, :] += bnd.T
    s[n - k:, n - k:] += np.diag(np.linspace(50, 500, k))
    s = (s + s.T) / 2
    _check_lowest(s, 13)


def test_potrf_trtri_norm_est(rng):
    a = rng.standard_normal((21, 21)); s = a @ a.T + 21 * np.eye(21)
    junk = np.triu(rng.standard_normal((21, 21)), 1)
    l, info = capi.potrf_lower(np.tril(s) + junk)           # the strict upper triangle is not referenced
    assert info == 0 and np.abs(np.tril(l) - np.linalg.cholesky(s)).max() < 1e-12
    assert np.array_equal(np.triu(l, 1), junk)
    li, info = capi.trtri_lower(l)
    assert info == 0 and np.abs(np.tril(li) - np.linalg.inv(np.linalg.cholesky(s))).max() < 1e-13
    assert np.array_equal(np.triu(li, 1), junk)
    lt = np.tril(l)
    want = np.abs(np.diag(lt)).max() + np.sqrt((np.tril(lt, -1) ** 2).sum())
    assert capi.norm_est(l) == pytest.approx(want, rel=1e-15)
    bad = s.copy(); bad[4, 4] = -3.0
    assert capi.potrf_lower(bad)[1] == 5                      # LAPACK's info = first failing column
    assert capi.potrf_lower(np.full((3, 3), np.nan))[1] == 1
