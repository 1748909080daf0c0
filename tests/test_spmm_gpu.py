"""GPU: the sample sparse operator -- ELLPACK SpMM with the reference's callback shape matvec(n,m,x,ax)
(reference README.md:34-35, callers main.f90:72-90) in device mode, and a whole Davidson / LOBPCG solve with it.

Oracle of the operator itself: scipy.sparse (exact arithmetic order differs: tolerance = the dot-product bound
64 eps |A| |x|).  Oracle of the solves: the oracle's drivers with the same matrix applied on the host (scipy callback),
and scipy.sparse.linalg.eigsh in shift-invert mode for the eigenvalues."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _laplacian_2d(nx, ny, shift=0.3):
    """5-point stencil plus a smooth diagonal: symmetric positive definite, 5 entries per row"""
    ex, ey = np.ones(nx), np.ones(ny)
    tx = sp.diags([-ex[:-1], 2 * ex, -ex[:-1]], [-1, 0, 1]); ty = sp.diags([-ey[:-1], 2 * ey, -ey[:-1]], [-1, 0, 1])
    a = sp.kron(sp.identity(ny), tx) + sp.kron(ty, sp.identity(nx))
    d = shift + 0.5 * np.sin(np.arange(nx * ny) * 0.01) ** 2
    return (a + sp.diags(d)).tocsr()


def _banded(n, half, rng):
    """symmetric band of half-width `half` with random entries and a dominant diagonal (2 half + 1 entries per row)"""
    diags = [rng.standard_normal(n - k) * 0.1 for k in range(1, half + 1)]
    a = sp.diags(diags, list(range(1, half + 1)), shape=(n, n))
    return (a + a.T + sp.diags(np.arange(1.0, n + 1.0) * 0.01 + 2.0)).tocsr()


@pytest.mark.parametrize("kind,n,m", [("lap", 96 * 64, 13), ("lap", 301 * 7, 5), ("band3", 5000, 8), ("band9", 4097, 13),
                                      ("band20", 2000, 3)])
def test_ell_spmm_matches_scipy(ctx, rng, kind, n, m):
    if kind == "lap":
        nx = 96 if n == 96 * 64 else 301
        a = _laplacian_2d(nx, n // nx)
    else:
        a = _banded(n, int(kind[4:]), rng)
    x = np.asfortranarray(rng.standard_normal((n, m)))
    ctx.spmm_setup(a)
    px, pax = ctx.panel(x), ctx.panel(n, m)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ctx._chk(ctx.lib.dla_call_matvec(ctx.h, capi.fn_address("dla_spmm_matvec"), n, m, px.ptr, pax.ptr))
        got = pax.download()
        ppx = ctx.panel(n, m)
        ctx._chk(ctx.lib.dla_call_precnd(ctx.h, capi.fn_address("dla_spmm_precnd"), n, m, -1.25, px.ptr, ppx.ptr))
        gotp = ppx.download()
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    want = a @ x
    assert np.all(np.abs(got - want) <= 64 * EPS * (abs(a) @ np.abs(x)) + 1e-300)
    den = a.diagonal() - 1.25
    wantp = np.where(np.abs(den)[:, None] > 1e-5, x / den[:, None], x)
    assert np.abs(gotp - wantp).max() <= 4 * EPS * np.abs(wantp).max()


@pytest.mark.parametrize("solver", ["davidson", "lobpcg"])
def test_solve_with_the_sparse_operator_on_the_device(ctx, oracle, solver):
    """The whole solve stays in HBM: panels, sparse matrix, preconditioner.  Same solve with the matrix applied by scipy
    on the host through the oracle's driver, and the eigenvalues from scipy's shift-invert Lanczos."""
    # the reference's test matrix made sparse: diagonal i + 1 (main.f90:312), a band of 1 / (i + j) couplings (main.f90:314)
    n, t, m, half = 20000, 6, 11, 6
    idx = np.arange(1.0, n + 1.0)
    offs = [1.0 / (idx[:-k] + idx[k:]) for k in range(1, half + 1)]
    a = sp.diags(offs, list(range(1, half + 1)), shape=(n, n))
    a = (a + a.T + sp.diags(idx + 1.0)).tocsr()
    diag = a.diagonal()
    # (unit vectors would make the first residual block rank deficient on a banded matrix: A e_j lives on 13 rows)
    g = np.asfortranarray(np.random.default_rng(5).random((n, m)) - 0.5)
    g[200:] *= 1e-3                       # most of the weight on the low end of the diagonal
    ctx.spmm_setup(a)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ev = ctx.panel(g)
        mv, pc = capi.fn_address("dla_spmm_matvec"), capi.fn_address("dla_spmm_precnd")
        if solver == "davidson":
            eig, _, ok, info = ctx.davidson_driver(n, t, m, 500, 1e-9, 20, 0.0, mv, pc, ev)
        else:
            eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 500, 1e-9, 0.0, mv, pc, ev)
        vec = ev.download()
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    assert ok, info

    # the oracle's driver with the same operator applied by scipy on the host
    c_dp, c_ip = C.POINTER(C.c_double), C.POINTER(C.c_int)

    def h_mv(pn, pm, px, pax):
        k = pm[0]
        x = np.ctypeslib.as_array(px, (k, n)).T
        np.ctypeslib.as_array(pax, (k, n)).T[:, :] = a @ x

    def h_pc(pn, pm, pf, px, ppx):
        k = pm[0]
        x = np.ctypeslib.as_array(px, (k, n)).T
        den = diag + pf[0]
        np.ctypeslib.as_array(ppx, (k, n)).T[:, :] = np.where(np.abs(den)[:, None] > 1e-5, x / den[:, None], x)

    cmv = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp)(h_mv)
    cpc = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp, c_dp)(h_pc)
    amv, apc = C.cast(cmv, C.c_void_p).value, C.cast(cpc, C.c_void_p).value
    if solver == "davidson":
        eo, vo, oko, tr = oracle.davidson(n, t, m, 500, 1e-9, 20, 0.0, amv, apc, g)
    else:
        eo, vo, oko, tr = oracle.lobpcg(n, t, m, 500, 1e-9, 0.0, amv, apc, g)
    assert oko
    assert np.allclose(eig[:t], eo[:t], rtol=1e-9, atol=0)
    assert abs(info["iters"] - tr.iters) <= max(2, tr.iters // 10), (info, tr.iters)
    want = np.sort(spl.eigsh(a, k=t, sigma=0.0, which="LM", return_eigenvectors=False))
    assert np.allclose(eig[:t], want, rtol=1e-7, atol=0)
    x = vec[:, :t]
    assert np.abs(x.T @ x - np.eye(t)).max() < 1e-10
    assert np.linalg.norm(a @ x - x * eig[None, :t], axis=0).max() < 1e-6
