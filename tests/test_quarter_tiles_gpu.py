"""GPU parity of the quarter-tile kernel variants: blocks whose last 16-column tile holds at most 8 live columns
(17..24 and 33..40 columns -- the reference's n_max = 21 and 37 among them) form that tile with v_mfma_f64_4x4x4
instead of a padded 16x16x4 instruction.  Every kernel that has the variant, every remainder 1..8, even n (the
16-byte path the variant lives on) with and without a row tail, against numpy with the dot-product error bound
of tests/test_kernels_gpu.py; and the variant switched off (tuning knob 7) must give the same bits for the
row-direction products (same summation order, only the instruction shape differs)."""
import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
WIDTHS = [17, 20, 21, 24, 33, 36, 37, 40]
TUNE0 = 100


def _gb(ax, au):
    return 64 * EPS * (ax.T @ au) + 1e-300


@pytest.mark.parametrize("k", WIDTHS)
@pytest.mark.parametrize("n", [2, 18, 64, 2000, 4110])
def test_gram_shapes(ctx, rng, n, k):
    u = np.asfortranarray(rng.standard_normal((n, k)))
    pu = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 1e30), u, np.full((n, 1), 1e30)])))
    uu = pu.col(1, k)
    want = u.T @ u
    got = ctx.gram(uu, uu)                                  # a block against itself: narrow tile on the A side
    assert np.all(np.abs(got - want) <= _gb(np.abs(u), np.abs(u)))
    assert np.array_equal(got, got.T)
    for l in (13, 48, 111, 2 * k):                          # projection-style: narrow tile on the B side, 1 and 2 X passes
        x = np.asfortranarray(rng.standard_normal((n, l)))
        got = ctx.gram(ctx.panel(x), uu)
        assert np.all(np.abs(got - x.T @ u) <= _gb(np.abs(x), np.abs(u))), l
    x = np.asfortranarray(rng.standard_normal((n, k)))      # lower triangle of X^T U, l == k, different panels
    got = ctx.gram_lower(ctx.panel(x), uu)
    low = np.tril(np.ones((k, k), bool))
    assert np.all(np.abs(got - x.T @ u)[low] <= _gb(np.abs(x), np.abs(u))[low])


@pytest.mark.parametrize("k", WIDTHS)
@pytest.mark.parametrize("n,l", [(2000, 111), (4110, 50), (64, 7), (2, 40), (30, 111)])
def test_row_products(ctx, rng, n, l, k):
    x = np.asfortranarray(rng.standard_normal((n, l)))
    c = np.asfortranarray(rng.standard_normal((l, k)))
    u = np.asfortranarray(rng.standard_normal((n, k)))
    px = ctx.panel(x)
    bound = 64 * EPS * (np.abs(x) @ np.abs(c)) + 1e-300
    res = {}
    for knob in (0, 1):
        ctx.set_option(TUNE0 + 7, knob)
        pz = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 7.0), np.zeros((n, k)), np.full((n, 1), 7.0)])))
        ctx.panel_gemm(px, c, pz.col(1, k))
        z = pz.download()
        assert np.all(z[:, 0] == 7.0) and np.all(z[:, -1] == 7.0)          # neighbours of the output block untouched
        pu = ctx.panel(u)
        ctx.panel_update(px, c, pu)
        w = np.asfortranarray(np.tril(rng.standard_normal((k, k))) + 3 * np.eye(k)) if knob == 0 else res[0][3]
        pt = ctx.panel(u)
        ctx.trmm_linvt(pt, w)
        res[knob] = (z[:, 1:-1], pu.download(), pt.download(), w)
    ctx.set_option(TUNE0 + 7, 0)
    z, upd, tr, w = res[0]
    assert np.all(np.abs(z - x @ c) <= bound)
    assert np.all(np.abs(upd - (u - x @ c)) <= bound + 4 * EPS * np.abs(u))
    assert np.all(np.abs(tr - u @ w.T) <= 64 * EPS * (np.abs(u) @ np.abs(w.T)) + 1e-300)
    for a, b in zip(res[0][:3], res[1][:3]):
        assert np.array_equal(a, b)                                          # same bits with the full-tile instruction


@pytest.mark.parametrize("knob", [0, 1, 5])
@pytest.mark.parametrize("k", WIDTHS + [41, 48])
@pytest.mark.parametrize("n,m", [(2000, 74), (4110, 20), (1998, 111), (6, 30), (34, 74)])
def test_fused_sweeps(ctx, rng, n, m, k, knob):
    """knob 1: full tiles only; knob 5: 64-row wave tiles in the three-tile sweeps (default: 32 rows when two blocks fit a CU)"""
    x = np.asfortranarray(rng.standard_normal((n, m)))
    u = np.asfortranarray(rng.standard_normal((n, k)))

    def check(got_u, got_g, want_u, scale):
        bound = 64 * EPS * scale + 1e-300
        assert np.all(np.abs(got_u - want_u) <= bound)
        want_g = want_u.T @ want_u
        gb = 64 * EPS * (np.abs(want_u).T @ np.abs(want_u)) + 2 * (np.abs(want_u).T @ bound) + 1e-300
        assert np.all(np.abs(got_g - want_g) <= gb)
        assert np.array_equal(got_g, got_g.T)

    ctx.set_option(TUNE0 + 7, knob)
    try:
        w = np.asfortranarray(np.triu(rng.standard_normal((k, k))) + 2 * np.eye(k))
        pu = ctx.panel(u)
        g = ctx.trmm_gram(pu, w)
        check(pu.download(), g, u @ w, np.abs(u) @ np.abs(w))
        c = np.asfortranarray(rng.standard_normal((m, k)) * 0.1)
        pu = ctx.panel(u)
        g = ctx.update_gram(ctx.panel(x), c, pu)
        check(pu.download(), g, u - x @ c, np.abs(x) @ np.abs(c) + np.abs(u))
        cp = np.asfortranarray(np.vstack([-c, w]))
        big = ctx.panel(np.asfortranarray(np.hstack([x, u])))
        g = ctx.combo_gram(big.col(0, m), cp, big.col(m, k))
        xu = np.hstack([x, u])
        check(big.col(m, k).download(), g, xu @ cp, np.abs(xu) @ np.abs(cp))
        assert np.array_equal(big.col(0, m).download(), x)
    finally:
        ctx.set_option(TUNE0 + 7, 0)


@pytest.mark.parametrize("m", WIDTHS)
@pytest.mark.parametrize("n,l", [(2000, 111), (4110, 40), (10, 60)])
def test_ritz_step(ctx, rng, n, l, m):
    v = np.asfortranarray(rng.standard_normal((n, l)))
    av = np.asfortranarray(rng.standard_normal((n, l)))
    y = np.asfortranarray(rng.standard_normal((l, m)))
    eig = rng.standard_normal(m)
    n_res = m - 2
    skip = np.zeros(m, np.int32); skip[1] = 1
    outs = []
    for knob in (0, 1):
        ctx.set_option(TUNE0 + 7, knob)
        pe, pr, pa = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m)
        rn = ctx.ritz_residual(ctx.panel(v), ctx.panel(av), y, eig, n_res, skip, pe, pr, pa)
        outs.append((pe.download(), pr.download(), pa.download(), rn.copy()))
    ctx.set_option(TUNE0 + 7, 0)
    ev, raw = v @ y, av @ y
    r = raw.copy()
    for i in range(n_res):
        if not skip[i]:
            r[:, i] -= eig[i] * ev[:, i]
    tol = 64 * EPS * (np.abs(v) @ np.abs(y) * (1 + np.abs(eig)[None, :]) + np.abs(av) @ np.abs(y)) + 1e-300
    e0, r0, a0, rn0 = outs[0]
    assert np.all(np.abs(e0 - ev) <= tol) and np.all(np.abs(r0 - r) <= tol) and np.all(np.abs(a0 - raw) <= tol)
    for i in range(n_res):
        if not skip[i]:
            assert np.isclose(rn0[0, i], np.linalg.norm(r0[:, i]) / np.sqrt(n), rtol=1e-13)
            assert np.isclose(rn0[1, i], np.abs(r0[:, i]).max(), rtol=1e-15)
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("l", [49, 63, 64, 65, 80, 89, 96, 100, 111, 112])
@pytest.mark.parametrize("n", [2, 14, 30, 64, 2000, 4110])
def test_lower_triangle_of_two_panels_single_pass(ctx, rng, n, l):
    """S^T A S of LOBPCG: the lower triangle of X^T U for two different panels of 4..7 column tiles is formed in one pass
    (tile pairs above the diagonal compiled out); knob 8 = the multi-pass kernel it replaces."""
    x = np.asfortranarray(rng.standard_normal((n, l)))
    u = np.asfortranarray(rng.standard_normal((n, l)))
    px = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 1e30), x, np.full((n, 1), 1e30)])))
    pu = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 1e30), u, np.full((n, 1), 1e30)])))
    low = np.tril(np.ones((l, l), bool))
    want = x.T @ u
    for knob in (0, 8):
        ctx.set_option(TUNE0 + 7, knob)
        got = ctx.gram_lower(px.col(1, l), pu.col(1, l))
        ctx.set_option(TUNE0 + 7, 0)
        assert np.all(np.abs(got - want)[low] <= _gb(np.abs(x), np.abs(u))[low]), knob
        assert np.all(np.isfinite(got))
