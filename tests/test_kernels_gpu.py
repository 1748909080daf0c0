"""GPU parity: each HIP block-algebra kernel, through the C-ABI, against the oracle's
restatement of the BLAS call it replaces.

Floating point: the kernels sum in a different order than the oracle, so the bar is a relative
tolerance on float64 results: |got - want| <= 64 eps * (|X|^T |U|) elementwise, i.e. the
standard dot-product error bound with a small constant.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _bound(ax, au, n):
    return 64 * EPS * (ax.T @ au) + 1e-300


SHAPES = [  # n, l, k
    (257, 1, 1), (257, 13, 5), (1000, 13, 13), (1000, 26, 13), (1024, 16, 16), (1001, 39, 13),
    (4096, 100, 13), (5000, 260, 13), (3000, 420, 21), (2000, 111, 37), (2000, 111, 111), (640, 200, 70),
]


@pytest.mark.parametrize("n,l,k", SHAPES)
def test_gram(ctx, oracle, rng, n, l, k):
    x = np.asfortranarray(rng.standard_normal((n, l)))
    u = np.asfortranarray(rng.standard_normal((n, k)))
    got = ctx.gram(ctx.panel(x), ctx.panel(u))
    want = oracle.gemm_tn(x, u)
    assert np.all(np.abs(got - want) <= _bound(np.abs(x), np.abs(u), n))


@pytest.mark.parametrize("n", [2, 6, 14, 16, 18, 30, 32, 34, 46, 62, 64, 66, 1000, 4110])
@pytest.mark.parametrize("l,k", [(4, 13), (13, 13), (26, 13), (65, 13), (104, 13), (143, 13), (191, 13), (208, 13), (100, 21), (60, 37)])
def test_gram_even_n_tails(ctx, oracle, rng, n, l, k):
    """Even n takes the LDS-staged kernel (16- or 32-row wave tiles): every tail residue, fewer rows than one tile,
    one pass of up to 12 column tiles and two passes."""
    x = np.asfortranarray(rng.standard_normal((n, l)))
    u = np.asfortranarray(rng.standard_normal((n, k)))
    # guard columns on both sides: a kernel that runs past a column's end would pick these up
    xp = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 1e30), x, np.full((n, 1), 1e30)])))
    up = ctx.panel(np.asfortranarray(np.hstack([np.full((n, 1), 1e30), u, np.full((n, 1), 1e30)])))
    got = ctx.gram(xp.col(1, l), up.col(1, k))
    want = oracle.gemm_tn(x, u)
    assert np.all(np.abs(got - want) <= _bound(np.abs(x), np.abs(u), n))
    if l == k:
        got2 = ctx.gram_lower(xp.col(1, l), up.col(1, k))
        low = np.tril(np.ones((l, l), bool))
        assert np.all(np.abs(got2 - want)[low] <= _bound(np.abs(x), np.abs(u), n)[low])


def test_gram_self_and_column_views(ctx, oracle, rng):
    n, lda, k = 2000, 60, 13
    big = np.asfortranarray(rng.standard_normal((n, lda)))
    p = ctx.panel(big)
    got = ctx.gram(p.col(0, 26), p.col(26, k))          # projection-style call on views of one panel
    want = oracle.gemm_tn(big[:, :26], big[:, 26:26 + k])
    assert np.all(np.abs(got - want) <= _bound(np.abs(big[:, :26]), np.abs(big[:, 26:26 + k]), n))
    g = ctx.gram(p.col(5, k), p.col(5, k))
    assert np.allclose(g, g.T, rtol=0, atol=1e-10)


@pytest.mark.parametrize("n,l,k", SHAPES[:11])
def test_panel_gemm_and_update(ctx, oracle, rng, n, l, k):
    if k > 48 * 3:
        pytest.skip("wide")
    x = np.asfortranarray(rng.standard_normal((n, l)))
    c = np.asfortranarray(rng.standard_normal((l, k)))
    u = np.asfortranarray(rng.standard_normal((n, k)))
    px, pz, pu = ctx.panel(x), ctx.panel(n, k), ctx.panel(u)
    ctx.panel_gemm(px, c, pz)
    want = oracle.gemm_nn(x, c)
    bound = 64 * EPS * (np.abs(x) @ np.abs(c)) + 1e-300
    assert np.all(np.abs(pz.download() - want) <= bound)
    ctx.panel_update(px, c, pu)
    want2 = oracle.gemm_nn(x, c, alpha=-1.0, beta=1.0, z=u)
    assert np.all(np.abs(pu.download() - want2) <= bound + 4 * EPS * np.abs(u))


@pytest.mark.parametrize("n,k", [(257, 1), (257, 5), (1000, 13), (4098, 16), (3000, 21), (2001, 37)])
def test_trmm_linvt(ctx, rng, n, k):
    u = np.asfortranarray(rng.standard_normal((n, k)))
    linv = np.asfortranarray(np.tril(rng.standard_normal((k, k))) + 3 * np.eye(k))
    linv_dirty = linv + np.triu(rng.standard_normal((k, k)), 1)   # the strict upper part must be ignored
    pu = ctx.panel(u)
    ctx.trmm_linvt(pu, linv_dirty)
    want = u @ linv.T
    bound = 64 * EPS * (np.abs(u) @ np.abs(linv.T)) + 1e-300
    assert np.all(np.abs(pu.download() - want) <= bound)


@pytest.mark.parametrize("n,l,m,n_res", [(257, 13, 13, 8), (1000, 26, 13, 8), (1000, 160, 8, 4), (5000, 260, 13, 8),
                                         (2000, 111, 37, 37), (2001, 63, 21, 16)])
def test_ritz_residual(ctx, oracle, rng, n, l, m, n_res):
    v = np.asfortranarray(rng.standard_normal((n, l)))
    av = np.asfortranarray(rng.standard_normal((n, l)))
    y = np.asfortranarray(rng.standard_normal((l + 3, m)))     # ldy > l on purpose
    eig = rng.standard_normal(m)
    skip = np.zeros(m, np.int32)
    skip[1] = 1 if n_res > 1 else 0
    pe, pr, pa = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m)
    rn = ctx.ritz_residual(ctx.panel(v), ctx.panel(av), y, eig, n_res, skip, pe, pr, pa)
    ev_want = oracle.gemm_nn(v, y[:l])
    avy_want = oracle.gemm_nn(av, y[:l])
    r_want = avy_want.copy()
    b1 = 64 * EPS * (np.abs(v) @ np.abs(y[:l])) + 1e-300
    b2 = 64 * EPS * (np.abs(av) @ np.abs(y[:l])) + 1e-300
    for i in range(n_res):
        if not skip[i]:
            r_want[:, i] -= eig[i] * ev_want[:, i]
    assert np.all(np.abs(pe.download() - ev_want) <= b1)
    assert np.all(np.abs(pa.download() - avy_want) <= b2)
    r_got = pr.download()
    assert np.all(np.abs(r_got - r_want) <= b2 + np.abs(eig)[None, :] * b1 + 4 * EPS * np.abs(r_want))
    for i in range(n_res):
        if skip[i]:
            assert rn[0, i] == 0.0 and rn[1, i] == 0.0     # untouched, like the reference's `cycle`
        else:
            assert np.isclose(rn[0, i], np.linalg.norm(r_got[:, i]) / np.sqrt(n), rtol=1e-13)
            assert rn[1, i] == np.abs(r_got[:, i]).max()


def test_axpy_nrm2_random_fill(ctx, oracle, rng):
    n, m = 3001, 7
    x = np.asfortranarray(rng.standard_normal((n, m))); y = np.asfortranarray(rng.standard_normal((n, m)))
    px, py = ctx.panel(x), ctx.panel(y)
    ctx.axpy(0.37, px, py)
    assert np.all(np.abs(py.download() - (y + 0.37 * x)) <= 2 * EPS * (np.abs(y) + np.abs(0.37 * x)))   # fma vs mul+add
    assert np.isclose(ctx.nrm2(px), np.linalg.norm(x), rtol=1e-14)
    ctx.random_fill(px)
    got = px.download()
    for (i, j) in [(0, 0), (17, 3), (n - 1, m - 1)]:
        assert got[i, j] == oracle.u01(7, i + 1, j + 1)       # bit-exact: integer hash -> double
    assert 0.0 <= got.min() and got.max() < 1.0


def test_empty_and_degenerate(ctx):
    p = ctx.panel(np.zeros((64, 3)))
    assert ctx.gram(p, p).tolist() == np.zeros((3, 3)).tolist()
    assert ctx.nrm2(p) == 0.0


@pytest.mark.parametrize("n,l", [(1000, 13), (2000, 39), (3000, 111), (1001, 70)])
def test_gram_lower(ctx, oracle, rng, n, l):
    """S^T AS for a dsyev('l') consumer: the lower triangle (incl. the 16 x 16 diagonal blocks) is exact,
    blocks strictly above the block diagonal may be zero."""
    x = np.asfortranarray(rng.standard_normal((n, l))); u = np.asfortranarray(rng.standard_normal((n, l)))
    got = ctx.gram_lower(ctx.panel(x), ctx.panel(u))
    want = oracle.gemm_tn(x, u)
    bound = _bound(np.abs(x), np.abs(u), n)
    low = np.tril(np.ones((l, l), bool))
    assert np.all(np.abs(got - want)[low] <= bound[low])
    up = ~low
    assert np.all((np.abs(got - want)[up] <= bound[up]) | (got[up] == 0.0))


@pytest.mark.parametrize("n,m,k", [(3000, 40, 21), (2001, 111, 37), (4096, 100, 48), (1000, 26, 17)])
def test_ortho_vs_x_wide_blocks_contiguous(ctx, oracle, rng, n, m, k):
    """Blocks wider than 16 columns through the fused sweeps (TRMM+Gram, [X|U] sweep+Gram with 2-3 column tiles):
    U is the block that follows X in one panel, as in the drivers."""
    x = np.linalg.qr(rng.standard_normal((n, m)))[0]
    u = rng.standard_normal((n, k)) + x[:, : min(m, k)] @ rng.standard_normal((min(m, k), k)) * 3
    panel = ctx.panel(np.asfortranarray(np.hstack([x, u])))
    ctx.ortho_vs_x(panel.col(0, m), panel.col(m, k))
    got = panel.col(m, k).download()
    want, _, st = oracle.ortho_vs_x(np.asfortranarray(x), np.asfortranarray(u))
    assert st == 0
    assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
    assert np.abs(x.T @ got).max() < 50 * EPS
    assert np.abs(got - want).max() < 1e-11
    assert np.array_equal(panel.col(0, m).download(), np.asfortranarray(x))      # X untouched


def test_allocator_cache_and_trim(ctx):
    """Freed panels are cached for reuse; dla_trim hands them back (include/diaglib_amd.h)."""
    ctx.trim()
    p = ctx.panel(4096, 64)          # 2 MiB granule
    addr = p.ptr
    p.free()
    q = ctx.panel(4096, 64)
    assert q.ptr == addr             # the cached block came back
    q.free()
    assert ctx.trim() >= 4096 * 64 * 8
    assert ctx.trim() == 0


@pytest.mark.parametrize("n,l,m", [(2000, 500, 37), (1501, 420, 48), (1000, 700, 21), (1200, 130, 55), (900, 300, 100)])
def test_ritz_residual_deep_subspace(ctx, oracle, rng, n, l, m):
    """Wide block times deep subspace (Davidson with n_max = 37 and 20 blocks reaches L = 740): Y no longer fits the
    kernel's LDS copy and the engine goes through the chunked products; blocks wider than 48 columns go in 48-column
    pieces."""
    v = np.asfortranarray(rng.standard_normal((n, l)))
    av = np.asfortranarray(rng.standard_normal((n, l)))
    y = np.asfortranarray(rng.standard_normal((l, m)) / np.sqrt(l))
    eig = rng.standard_normal(m)
    n_res = m - 3
    skip = np.zeros(m, np.int32); skip[1] = 1
    pe, pr, pa = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m)
    rn = ctx.ritz_residual(ctx.panel(v), ctx.panel(av), y, eig, n_res, skip, pe, pr, pa)
    ev = v @ y
    r = av @ y
    raw = r.copy()
    for i in range(n_res):
        if not skip[i]:
            r[:, i] -= eig[i] * ev[:, i]
    tol = 64 * EPS * (np.abs(v) @ np.abs(y) * (1 + np.abs(eig)[None, :]) + np.abs(av) @ np.abs(y)) + 1e-300
    assert np.all(np.abs(pe.download() - ev) <= tol)
    assert np.all(np.abs(pr.download() - r) <= tol)
    assert np.all(np.abs(pa.download() - raw) <= tol)
    for i in range(n_res):
        if not skip[i]:
            assert np.isclose(rn[0, i], np.linalg.norm(r[:, i]) / np.sqrt(n), rtol=1e-10)
            assert np.isclose(rn[1, i], np.abs(r[:, i]).max(), rtol=1e-10)


@pytest.mark.parametrize("n,m,k", [(257, 13, 5), (1000, 26, 13), (4096, 104, 13), (3000, 63, 21), (2001, 74, 37), (2000, 111, 37),
                                   (640, 16, 16), (5000, 247, 13)])
def test_fused_sweeps_against_their_blas_pairs(ctx, oracle, rng, n, m, k):
    """the fused epilogues checked directly, not only through ortho_vs_x.  Each fused sweep against the
    oracle's dgemm('n','n') for the update and dgemm('t','n') for the Gram matrix of the updated block
    (reference diaglib.f90:3327+3256, 3544+3256)."""
    x = np.asfortranarray(rng.standard_normal((n, m)))
    u = np.asfortranarray(rng.standard_normal((n, k)))

    def check(got_u, got_g, want_u, scale):
        bound = 64 * EPS * scale + 1e-300
        assert np.all(np.abs(got_u - want_u) <= bound)
        want_g = oracle.gemm_tn(want_u, want_u)
        gb = 64 * EPS * (np.abs(want_u).T @ np.abs(want_u)) + 2 * (np.abs(want_u).T @ bound) + 1e-300
        assert np.all(np.abs(got_g - want_g) <= gb)
        assert np.array_equal(got_g, got_g.T)                       # complete and symmetric

    # U <- U W with a general (upper triangular here) W
    w = np.asfortranarray(np.triu(rng.standard_normal((k, k))) + 2 * np.eye(k))
    pu = ctx.panel(u)
    g = ctx.trmm_gram(pu, w)
    check(pu.download(), g, oracle.gemm_nn(u, w), np.abs(u) @ np.abs(w))
    # U <- U - X C
    c = np.asfortranarray(rng.standard_normal((m, k)) * 0.1)
    pu = ctx.panel(u)
    g = ctx.update_gram(ctx.panel(x), c, pu)
    check(pu.download(), g, oracle.gemm_nn(x, c, alpha=-1.0, beta=1.0, z=u), np.abs(x) @ np.abs(c) + np.abs(u))
    # U <- [X | U] C' on one contiguous panel
    cp = np.asfortranarray(np.vstack([-c, w]))
    big = ctx.panel(np.asfortranarray(np.hstack([x, u])))
    g = ctx.combo_gram(big.col(0, m), cp, big.col(m, k))
    xu = np.asfortranarray(np.hstack([x, u]))
    check(big.col(m, k).download(), g, oracle.gemm_nn(xu, cp), np.abs(xu) @ np.abs(cp))
    assert np.array_equal(big.col(0, m).download(), x)              # X untouched


@pytest.mark.parametrize("n,l,m,k2", [(2000, 39, 13, 13), (4110, 78, 13, 9), (2000, 111, 37, 37), (2001, 111, 37, 30), (3000, 63, 21, 21),
                                      (1000, 26, 13, 1), (64, 111, 37, 37), (2000, 260, 48, 48), (2000, 111, 37, 0)])
def test_ritz_sweep_with_extra_products(ctx, rng, n, l, m, k2):
    """LOBPCG's new P block (reference diaglib.f90:495-501: P = S cp, AP = AS cp) formed by the Ritz sweep itself: the same
    pass over V and AV yields evec, r, AV Y, the norms AND V C2, AV C2.  Everything must equal the separate calls bit for bit
    (same contraction order), also where the engine falls back to them (odd n, more than five column tiles)."""
    v = np.asfortranarray(rng.standard_normal((n, l)))
    av = np.asfortranarray(rng.standard_normal((n, l)))
    y = np.asfortranarray(rng.standard_normal((l + 2, m)))
    c2 = np.asfortranarray(rng.standard_normal((l + 1, k2))) if k2 else np.zeros((l, 0), order="F")
    eig = rng.standard_normal(m)
    n_res = m
    skip = np.zeros(m, np.int32); skip[m // 2] = 1
    pv, pav = ctx.panel(v), ctx.panel(av)
    pe, pr, pa = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m)
    rn_ref = ctx.ritz_residual(pv, pav, y, eig, n_res, skip, pe, pr, pa)
    ref = (pe.download(), pr.download(), pa.download())
    pe2, pr2, pa2 = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m)
    pp = ctx.panel(np.full((n, max(k2, 1) + 2), 7.0, order="F")); pap = ctx.panel(np.full((n, max(k2, 1) + 2), 7.0, order="F"))
    rn = ctx.ritz_residual_p(pv, pav, y, eig, n_res, skip, pe2, pr2, pa2, c2[:l] if k2 else c2, pp.col(1, max(k2, 1)), pap.col(1, max(k2, 1)))
    for a, b in zip(ref, (pe2.download(), pr2.download(), pa2.download())):
        assert np.array_equal(a, b)
    assert np.array_equal(rn, rn_ref)
    gp, gap = pp.download(), pap.download()
    assert np.all(gp[:, 0] == 7.0) and np.all(gp[:, k2 + 1:] == 7.0) and np.all(gap[:, 0] == 7.0) and np.all(gap[:, k2 + 1:] == 7.0)
    if k2:
        zp, zap = ctx.panel(n, k2), ctx.panel(n, k2)
        ctx.panel_gemm(pv, c2[:l], zp); ctx.panel_gemm(pav, c2[:l], zap)
        assert np.array_equal(gp[:, 1:k2 + 1], zp.download())
        assert np.array_equal(gap[:, 1:k2 + 1], zap.download())
        bound = 64 * EPS * (np.abs(v) @ np.abs(c2[:l])) + 1e-300
        assert np.all(np.abs(gp[:, 1:k2 + 1] - v @ c2[:l]) <= bound)


@pytest.mark.parametrize("n,l,m", [(4096, 40, 8), (3001, 77, 13), (2048, 160, 8), (2500, 64, 21), (1024, 30, 52)])
def test_ritz_sweep_with_two_coefficient_blocks(ctx, rng, n, l, m):
    """dla_ritz_residual2: e = V y1, r = AV y2 - theta e and the norms of r in one sweep (the residual blocks of the linear-response
    drivers, reference diaglib.f90:872-889, 1337-1353: two dgemms + daxpy / dnrm2), against numpy; skipped roots stay uncorrected;
    m = 52 goes through the three-sweep default."""
    v = np.asfortranarray(rng.standard_normal((n, l))); av = np.asfortranarray(rng.standard_normal((n, l)))
    y1 = np.asfortranarray(rng.standard_normal((l, m))); y2 = np.asfortranarray(rng.standard_normal((l, m)))
    theta = rng.standard_normal(m)
    n_res = max(1, m - 2)
    skip = np.zeros(n_res, np.int32); skip[0] = 1
    pv, pav, pe, pr = ctx.panel(v), ctx.panel(av), ctx.panel(n, m), ctx.panel(n, m)
    rn = ctx.ritz_residual2(pv, pav, y1, y2, theta, n_res, skip, pe, pr)
    e_want = v @ y1
    r_want = av @ y2
    for j in range(n_res):
        if not skip[j]:
            r_want[:, j] -= theta[j] * e_want[:, j]
    bound = 64 * EPS * (np.abs(v) @ np.abs(y1) + np.abs(av) @ np.abs(y2) * (1 + np.abs(theta).max()))
    assert np.all(np.abs(pe.download() - e_want) <= bound)
    assert np.all(np.abs(pr.download() - r_want) <= bound)
    for j in range(n_res):
        if skip[j]:
            continue
        assert np.isclose(rn[0, j], np.linalg.norm(r_want[:, j]) / np.sqrt(n), rtol=1e-12)
        assert np.isclose(rn[1, j], np.abs(r_want[:, j]).max(), rtol=1e-12)
