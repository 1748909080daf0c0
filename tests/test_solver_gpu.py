"""GPU parity: orthogonalisation kernels and the two drivers, through the C-ABI / Fortran
drivers, against the oracle on the same seeded inputs.

Tolerances (float64, different summation order than the oracle):
  * orthonormality / orthogonality to X:  <= 50 eps  (the reference's own stopping rule is
    growth*eps < 2 eps, diaglib.f90:3562-3564);
  * well-conditioned panels agree entry-wise to 1e-12; for ill-conditioned inputs only the
    invariants (span, orthonormality) are compared -- Cholesky-QR amplifies rounding by cond(U)^2;
  * eigenvalues: relative 1e-11; per-iteration traces (rms, max): relative 1e-6 above 1e-12 floor
    -- iteration counts must match exactly.
"""
import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _panel_with_cond(rng, n, k, cond):
    q, _ = np.linalg.qr(rng.standard_normal((n, k)))
    s = np.logspace(0, -np.log10(cond), k) if k > 1 else np.ones(1)
    w, _ = np.linalg.qr(rng.standard_normal((k, k)))
    return np.asfortranarray((q * s) @ w.T)


@pytest.mark.parametrize("n,k,cond", [(257, 1, 1), (257, 5, 1e2), (1000, 13, 1e0), (1000, 13, 1e6), (2000, 13, 1e12),
                                      (3000, 21, 1e3), (2000, 37, 1e4)])
def test_ortho_cd(ctx, oracle, rng, n, k, cond):
    u = _panel_with_cond(rng, n, k, cond)
    p = ctx.panel(u)
    growth, ok = ctx.ortho_cd(p)
    got = p.download()
    want, g_want, ok_want, n_macro = oracle.ortho_cd(u)
    assert ok and ok_want
    assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
    if cond <= 1e3:
        assert np.abs(got - want).max() < 1e-12
        assert np.isclose(growth, g_want, rtol=1e-10)
    else:
        # same span: projector difference small relative to conditioning
        assert np.abs(got @ (got.T @ want) - want).max() < max(1e-6, 100 * cond * EPS)
        assert np.isclose(np.log10(growth), np.log10(g_want), atol=0.5)


def test_ortho_cd_rank_deficient_takes_shift_ladder(ctx, oracle, rng):
    n, k = 1000, 6
    u = np.asfortranarray(rng.standard_normal((n, k)))
    u[:, 5] = u[:, 0] + u[:, 1]          # exactly dependent column -> Cholesky fails -> level shift (3265-3295)
    p = ctx.panel(u)
    growth, ok = ctx.ortho_cd(p)
    got = p.download()
    assert ok
    assert np.all(np.isfinite(got))
    assert np.abs(got[:, :5].T @ got[:, :5] - np.eye(5)).max() < 1e-10


@pytest.mark.parametrize("n,m,k", [(257, 13, 13), (1000, 39, 13), (1001, 26, 5), (4000, 260, 13), (3000, 111, 37),
                                   (500, 0, 4)])
def test_ortho_vs_x(ctx, oracle, rng, n, m, k):
    x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, max(m, 1))))[0][:, :m])
    u = np.asfortranarray(rng.standard_normal((n, k)) + (x[:, : min(m, k)] @ rng.standard_normal((min(m, k), k)) * 5 if m else 0))
    px, pu = ctx.panel(x if m else np.zeros((n, 1))), ctx.panel(u)
    ctx.ortho_vs_x(px, pu, m=m)
    got = pu.download()
    want, n_outer, st = oracle.ortho_vs_x(x, u)
    assert st == 0
    assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
    if m:
        assert np.abs(x.T @ got).max() < 50 * EPS
    assert np.abs(got - want).max() < 1e-11


def test_b_ortho_and_b_ortho_vs_x(ctx, oracle, rng):
    n, m, k = 600, 8, 5
    a = rng.standard_normal((n, n)) * 0.01
    b = a @ a.T + np.eye(n)
    x = rng.standard_normal((n, m))
    lx = np.linalg.cholesky(x.T @ b @ x)
    x = np.asfortranarray(x @ np.linalg.inv(lx).T)          # B-orthonormal X
    bx = np.asfortranarray(b @ x)
    u = np.asfortranarray(rng.standard_normal((n, k)))
    pu = ctx.panel(u)
    ctx.b_ortho_vs_x(ctx.panel(x), ctx.panel(bx), pu)
    got = pu.download()
    want, st = oracle.b_ortho_vs_x(x, bx, u)
    assert st == 0 and np.abs(got - want).max() < 1e-11
    assert np.abs(bx.T @ got).max() < 1e-13
    bu = np.asfortranarray(b @ got)
    pu2, pbu = ctx.panel(got), ctx.panel(bu)
    ctx.b_ortho(pu2, pbu)
    g2, gb2 = pu2.download(), pbu.download()
    w2, wb2 = oracle.b_ortho(got, bu)
    assert np.abs(g2 - w2).max() < 1e-12 and np.abs(gb2 - wb2).max() < 1e-12
    assert np.abs(g2.T @ gb2 - np.eye(k)).max() < 1e-13


def test_check_guess_paths(ctx, oracle, rng):
    n, m = 1200, 6
    # (a) zero guess -> documented generator + ortho_cd, identical stream to the oracle
    p = ctx.panel(np.zeros((n, m)))
    ctx.check_guess(p)
    got = p.download()
    want = oracle.check_guess(np.zeros((n, m), order="F"))
    assert np.abs(got - want).max() < 1e-12
    # (b) unit vectors are exactly orthonormal -> untouched (exact-equality test, diaglib.f90:3774)
    e = np.zeros((n, m), order="F"); e[np.arange(m), np.arange(m)] = 1.0
    p = ctx.panel(e); ctx.check_guess(p)
    assert np.array_equal(p.download(), e)
    # (c) generic guess -> orthonormalised
    g = np.asfortranarray(rng.random((n, m)) - 0.5)
    p = ctx.panel(g); ctx.check_guess(p)
    got = p.download()
    assert np.abs(got - oracle.check_guess(g)).max() < 1e-12


def test_get_coeffs(ctx, oracle, rng):
    n_max, n_act = 6, 4
    len_u = n_max + 2 * n_act
    len_a = 3 * n_max
    q, _ = np.linalg.qr(rng.standard_normal((len_u, len_u)))
    a_red = np.zeros((len_a, len_a), order="F"); a_red[:len_u, :len_u] = q
    ux, up = ctx.get_coeffs(a_red, len_u, n_max, n_act)
    wx, wp = oracle.get_coeffs(a_red, len_u, n_max, n_act)
    assert np.array_equal(ux, wx)
    assert np.abs(up - wp).max() < 1e-13


# ----------------------------------------------------------------------------- drivers
def _cmp_trace(info, tr, eig, eo, n_targ, exact=True):
    """Iteration counts: exact for well-separated convergence histories.  With a random guess the
    locking decisions sit on the tol threshold and LOBPCG's path depends on last-bit differences of
    the Gram sums: the reference itself (flang+MKL) takes 45 iterations where the oracle takes 43 on
    the n=2000 case below and the HIP path has been seen between 46 and 51 across kernel revisions that differ
    only in summation order (parity notes), so those cases allow 15 %."""
    if exact:
        assert info["iters"] == tr.iters, (info, tr.iters)
        assert info["matvec_cols"] == tr.matvec_cols, (info, tr.matvec_cols)
    else:
        assert abs(info["iters"] - tr.iters) <= max(1, (15 * tr.iters) // 100), (info, tr.iters)
        assert abs(info["matvec_cols"] - tr.matvec_cols) <= max(8, (15 * tr.matvec_cols) // 100), (info, tr.matvec_cols)
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-11, atol=0)


def _cmp_vecs(v, vo, n_targ, tol):
    sgn = np.sign((v * vo).sum(0))
    assert np.abs(v * sgn - vo)[:, :n_targ].max() < tol


@pytest.mark.parametrize("guess_kind", ["unit", "random"])
@pytest.mark.parametrize("solver", ["davidson", "lobpcg"])
def test_dense_reference_matrix_host_callbacks(ctx, oracle, rng, solver, guess_kind):
    """BASELINE cfg 1 sizing: dense a_ii=i+1, a_ij=1/(i+j) (main.f90:311-317), n=2000, 4 roots, n_max=8,
    max_dav=20 -- drop-in mode: HOST callbacks (the oracle's C operator) staged by the engine."""
    n, n_targ, n_max = 2000, 4, 8
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    if guess_kind == "unit":
        g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    else:
        g = np.asfortranarray(rng.random((n, n_max)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if solver == "davidson":
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
        assert info["restarts"] == tr.restarts
    else:
        eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
    assert ok and oko
    _cmp_trace(info, tr, eig, eo, n_targ, exact=(guess_kind == "unit"))
    _cmp_vecs(v, vo, n_targ, 1e-6)
    # LAPACK-style cross-check of main.f90:321-342
    a = 1.0 / (np.arange(1, n + 1)[:, None] + np.arange(1, n + 1)[None, :]); np.fill_diagonal(a, np.arange(1, n + 1) + 1.0)
    assert np.allclose(eig[:n_targ], np.linalg.eigvalsh(a)[:n_targ], rtol=0, atol=1e-7)


def test_host_callbacks_see_every_block_once_by_default(ctx, oracle):
    """Drop-in default: the caller's matvec / precnd are called once per block with the whole block, like the reference
    (diaglib.f90:1685, 1786) -- as many matvec calls as sweeps, every column exactly once; column chunks only on request."""
    import ctypes as C
    n, n_targ, n_max = 3000, 6, 11
    oracle.dense_setup(n)
    mv_addr, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    inner = C.cast(mv_addr, capi.MATVEC_T)
    calls = []

    def counted(pn, pm, px, pax):
        calls.append(pm[0])
        inner(pn, pm, px, pax)

    cb = capi.MATVEC_T(counted)
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, C.cast(cb, C.c_void_p).value, pc, g)
    assert ok and len(calls) == info["iters"] and sum(calls) == info["matvec_cols"], (calls, info)
    ctx.set_option(capi.OPT_STAGE_CHUNKS, 2)
    try:
        calls.clear()
        eig2, v2, ok2, info2 = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, C.cast(cb, C.c_void_p).value, pc, g)
    finally:
        ctx.set_option(capi.OPT_STAGE_CHUNKS, 0)
    assert ok2 and len(calls) > info2["iters"] and sum(calls) == info2["matvec_cols"] and info2["iters"] == info["iters"]
    assert np.allclose(eig2[:n_targ], eig[:n_targ], rtol=1e-12, atol=0)


@pytest.mark.parametrize("chunks", [2, 3, 5])
def test_host_callbacks_in_column_chunks_overlap_the_projection(ctx, oracle, chunks):
    """Drop-in mode with the block cut into column chunks (DLA_OPT_STAGE_CHUNKS >= 2, on request): every
    chunk flows download | caller's routine | upload, and the projection sweep of a chunk (its columns of [X | U]^T AU,
    reference diaglib.f90:1691) runs behind its upload while the caller's routine has the next chunk.
    Same history as the oracle, eigenvalues to 1e-12, and as many calls of the caller's routine as chunks."""
    n, n_targ, n_max = 3000, 6, 11
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    ctx.set_option(capi.OPT_STAGE_CHUNKS, chunks)
    try:
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
    finally:
        ctx.set_option(capi.OPT_STAGE_CHUNKS, 0)
    eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
    assert ok and oko and info["iters"] == tr.iters and info["matvec_cols"] == tr.matvec_cols
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-12, atol=0)
    _cmp_vecs(v, vo, n_targ, 1e-6)


def test_davidson_python_callbacks_and_restarts(ctx, oracle, rng):
    """max_dav=10 forces several restarts; callbacks are Python callables."""
    n, n_targ, n_max = 1000, 10, 15
    idx = np.arange(1, n + 1, dtype=np.float64)
    a = 1.0 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1.0)
    d = np.diag(a).copy()
    g = np.asfortranarray(rng.random((n, n_max)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eig, v, ok, info = ctx.davidson_driver(
        n, n_targ, n_max, 200, 1e-8, 5, 0.0, lambda x: a @ x,
        lambda fac, x: np.where(np.abs(d + fac)[:, None] > 1e-5, x / (d + fac)[:, None], x), g)
    oracle.dense_setup(n)
    eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 200, 1e-8, 5, 0.0, oracle.fn("orc_dense_matvec"),
                                      oracle.fn("orc_dense_precnd"), g)
    assert ok and oko and tr.restarts >= 1
    assert info["restarts"] == tr.restarts
    assert abs(info["iters"] - tr.iters) <= 1          # python matvec (BLAS order) vs C operator: last-bit differences
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-10, atol=0)


@pytest.mark.parametrize("solver", ["davidson", "lobpcg"])
def test_synthetic_operator_device_callbacks(ctx, oracle, solver):
    """matrix-free D + sigma W W^T at n=1e5, 8 roots, n_max=13, unit-vector guess,
    device-resident callbacks and evec."""
    n, n_targ, n_max = 100000, 8, 13
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.synth_setup(n, 0, n)
    oracle.synth_setup(n, 0, n)
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ev = ctx.panel(g)
    mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
    omv, opc = oracle.fn("orc_synth_matvec"), oracle.fn("orc_synth_precnd")
    try:
        if solver == "davidson":
            eig, _, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, ev)
            eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, omv, opc, g)
        else:
            eig, _, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, ev)
            eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.0, omv, opc, g)
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    assert ok and oko
    # rank-4 operator + unit guess: the 13 preconditioned residuals span only ~4 new directions, the
    # other columns of each new block are amplified rounding noise (ortho_cd goes through its
    # level-shift ladder), so residual histories differ at the 1-10 % level between ANY two
    # implementations -- reference (flang+MKL) vs oracle included (tools/trace_compare.py).
    _cmp_trace(info, tr, eig, eo, n_targ, exact=False)
    _cmp_vecs(ev.download(), vo, n_targ, 1e-6)
    # the survey's measured eigenvalues for this operator
    want = [2.862448, 3.655288, 4.438022, 5.252332, 6.194038, 7.016243, 8.211559, 9.146675]
    if False:
        assert np.allclose(eig[:8], want, atol=1e-5)   # survey used a different W generator (sin), kept for the record


def test_synth_matvec_matches_oracle(ctx, oracle, rng):
    n, m = 50001, 13
    ctx.synth_setup(n, 0, n); oracle.synth_setup(n, 0, n)
    x = np.asfortranarray(rng.standard_normal((n, m)))
    px, pax = ctx.panel(x), ctx.panel(n, m)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ctx.synth_matvec(px, pax)
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    w = oracle.synth_w()
    d = np.arange(1, n + 1) + 1.0
    want = d[:, None] * x + 0.5 * (w @ (w.T @ x))
    assert np.abs(pax.download() - want).max() < 1e-9 * np.abs(want).max()


def test_lobpcg_shift(ctx, oracle, rng):
    n, n_targ, n_max = 1500, 3, 6
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.asfortranarray(rng.random((n, n_max)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.75, mv, pc, g)
    eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.75, mv, pc, g)
    assert ok and oko
    _cmp_trace(info, tr, eig, eo, n_targ, exact=False)   # returned eig includes the shift (diaglib.f90:416, App. B 1)


def test_rccl_communicator_single_rank(ctx, oracle, rng):
    """The RCCL door on one GPU: a 1-rank communicator routes every reduction through ncclAllReduce on the
    engine's stream (N > 1 needs N GPUs; the sharded arithmetic itself is covered by tests/test_hostsim.py)."""
    x = np.asfortranarray(rng.standard_normal((4000, 20))); u = np.asfortranarray(rng.standard_normal((4000, 7)))
    want = ctx.gram(ctx.panel(x), ctx.panel(u))
    before = ctx.stats()["allreduces"]
    ctx.comm_init(1, 0, ctx.unique_id())
    try:
        ctx.set_shard(4000, 0)
        got = ctx.gram(ctx.panel(x), ctx.panel(u))
        assert np.array_equal(got, want)
        assert ctx.stats()["allreduces"] > before
        n, t, m = 3000, 3, 6
        oracle.dense_setup(n)
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        ctx.set_shard(n, 0)
        eig, v, ok, info = ctx.davidson_driver(n, t, m, 100, 1e-8, 20, 0.0, oracle.fn("orc_dense_matvec"),
                                               oracle.fn("orc_dense_precnd"), g)
        eo, vo, oko, tr = oracle.davidson(n, t, m, 100, 1e-8, 20, 0.0, oracle.fn("orc_dense_matvec"),
                                          oracle.fn("orc_dense_precnd"), g)
        assert ok and oko and info["iters"] == tr.iters and np.allclose(eig[:t], eo[:t], rtol=1e-11)
        eig, v, ok, info = ctx.lobpcg_driver(n, t, m, 100, 1e-8, 0.0, oracle.fn("orc_dense_matvec"),
                                             oracle.fn("orc_dense_precnd"), g)
        assert ok and np.allclose(eig[:t], eo[:t], rtol=1e-9)
        # device-driven chain under RCCL with a plan that does not fit (launches whose turn it is not still run their
        # collective -- out of place, so that nothing is scaled or consumed): a random block first (sets the plan), then an
        # already orthonormal one (fewer steps than planned), both against the oracle
        nn, mm, kk = 3000, 40, 11
        q = np.linalg.qr(rng.standard_normal((nn, mm + kk)))[0]
        xo = np.asfortranarray(q[:, :mm])
        ctx.set_shard(nn, 0)
        for u0 in (np.asfortranarray(rng.standard_normal((nn, kk))), np.asfortranarray(q[:, mm:])):
            big = ctx.panel(np.asfortranarray(np.hstack([xo, u0])))
            px, pu = big.col(0, mm), big.col(mm, kk)
            ctx.ortho_vs_x(px, pu)
            got = pu.download()
            want = oracle.ortho_vs_x(xo, u0)[0]
            assert np.abs(got - want).max() < 1e-11
            assert np.abs(xo.T @ got).max() < 50 * np.finfo(float).eps
    finally:
        ctx.comm_finalize()
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("solver,n,n_targ,n_max", [("davidson", 3001, 4, 8),      # odd n: 8-byte access path (VEC=1)
                                                   ("lobpcg", 2001, 4, 8),
                                                   ("davidson", 3000, 16, 21),    # BASELINE cfg 4 block width (2 column tiles)
                                                   ("lobpcg", 3000, 16, 21),
                                                   ("davidson", 2500, 32, 37),    # BASELINE cfg 5 block width (3 column tiles)
                                                   ("lobpcg", 2500, 32, 37)])
def test_block_widths_and_odd_n(ctx, oracle, solver, n, n_targ, n_max):
    """Shapes of BASELINE configs 4 and 5 (n_max = 21, 37: multi-tile kernels, unfused ortho path, LOBPCG's
    111 x 111 projected matrix) and odd row counts, on the reference's dense test matrix with a unit guess."""
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if solver == "davidson":
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
    else:
        eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
    assert ok and oko
    # Iteration counts and matvec columns are the oracle's, with ONE case decided at a tolerance edge.  The new blocks of these
    # runs are numerically rank deficient (cond(U) ~ 1e15 from the unit guess on this matrix: the level-shift ladder runs), so
    # their weakest columns are amplified rounding noise and the residuals of the last roots move by a few per cent with the
    # order of operations.  LOBPCG at n_max = 37: the oracle's root 32 stands at rms 1.09e-8 in iteration 4 (tol 1e-8), the
    # default schedule (one factorisation step in front of the loop, X^T U and U^T U in one sweep) at 0.83e-8 -- it stops after
    # 4 iterations / 127 columns where the oracle takes 5 / 133 (measured r04, tools/iters_probe.py).  With the reference's
    # order of operations (tune knob 6 = 7: leading ortho_cd iterated to convergence, separate sweeps) every case reproduces
    # the oracle's count exactly: test_block_widths_reference_order below.
    edge = (solver, n_max) == ("lobpcg", 37)
    if edge:
        assert info["iters"] in (tr.iters - 1, tr.iters) and abs(info["matvec_cols"] - tr.matvec_cols) <= 6
    else:
        assert info["iters"] == tr.iters
        assert abs(info["matvec_cols"] - tr.matvec_cols) <= 1          # (93 / 92 at n_max = 37: one root locks a sweep later)
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-11, atol=0)
    _cmp_vecs(v, vo, n_targ, 1e-6)


@pytest.mark.parametrize("solver,n,n_targ,n_max", [("davidson", 3000, 16, 21), ("lobpcg", 3000, 16, 21),
                                                   ("davidson", 2500, 32, 37), ("lobpcg", 2500, 32, 37)])
def test_block_widths_reference_order(ctx, oracle, solver, n, n_targ, n_max):
    """with the leading ortho_cd iterated to convergence and separate X^T U / U^T U sweeps (the reference's order
    of operations, diaglib.f90:3533-3568; tune knob 6 = 7) the device-driven chains reproduce the oracle's iteration counts
    EXACTLY, the n_max = 37 edge case included -- a real regression cannot hide in the default schedule's one-iteration band."""
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    ctx.set_option(100 + 6, 7)
    try:
        if solver == "davidson":
            eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
            eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
        else:
            eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
            eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
    finally:
        ctx.set_option(100 + 6, 0)
    assert ok and oko
    assert info["iters"] == tr.iters and abs(info["matvec_cols"] - tr.matvec_cols) <= 1, (info, tr.iters, tr.matvec_cols)
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-11, atol=0)


def test_davidson_restarts_with_device_callbacks(ctx, oracle, rng):
    """Restart path (diaglib.f90:1795-1825 incl. the n_rst zero-column quirk) with device-resident callbacks:
    12 of 13 roots wanted, max_dav=10 -> the basis fills up and restarts once with locked roots."""
    n, t, m = 40000, 12, 13
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ctx.synth_setup(n, 0, n); oracle.synth_setup(n, 0, n)
        ev = ctx.panel(g)
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 400, 1e-13, 10, 0.0, capi.fn_address("dla_synth_matvec"),
                                               capi.fn_address("dla_synth_precnd"), ev)
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eo, vo, oko, tr = oracle.davidson(n, t, m, 400, 1e-13, 10, 0.0, oracle.fn("orc_synth_matvec"),
                                      oracle.fn("orc_synth_precnd"), g)
    assert ok and oko and info["restarts"] >= 1 and tr.restarts >= 1, (ok, oko, info, tr.iters, tr.restarts)
    # tol=1e-13 sits close to the rounding floor of the 12th root: the tail of the history (how many 2-column
    # iterations the last root needs) differs between implementations, so only the outcome is compared
    assert abs(info["iters"] - tr.iters) <= max(2, tr.iters // 3)
    assert np.allclose(eig[:t], eo[:t], rtol=1e-11, atol=0)
    x = ev.download()[:, :t]
    assert np.abs(x.T @ x - np.eye(t)).max() < 1e-12
    _cmp_vecs(ev.download(), vo, t, 1e-6)


@pytest.mark.parametrize("solver", ["davidson", "lobpcg"])
@pytest.mark.parametrize("n_targ,n_max", [(1, 1), (3, 3), (1, 6)])
def test_driver_block_edge_cases(ctx, oracle, solver, n_targ, n_max):
    """No guard vectors (n_max == n_targ), a single vector, one wanted root in a wide block: reference dense
    matrix, unit-vector guess, against the oracle."""
    n = 800
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if solver == "davidson":
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 200, 1e-9, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 200, 1e-9, 20, 0.0, mv, pc, g)
    else:
        eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 200, 1e-9, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 200, 1e-9, 0.0, mv, pc, g)
    assert ok and oko
    _cmp_trace(info, tr, eig, eo, n_targ, exact=False)
    _cmp_vecs(v, vo, n_targ, 1e-6)


@pytest.mark.parametrize("solver", ["davidson", "lobpcg"])
def test_driver_gives_up_at_max_iter_like_the_reference(ctx, oracle, solver):
    """max_iter too small: ok = false, eig/evec hold the current Ritz pairs (SURVEY 8b ownership row)."""
    n, n_targ, n_max = 1000, 4, 8
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.asfortranarray(np.random.default_rng(4).random((n, n_max)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if solver == "davidson":
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 4, 1e-10, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 4, 1e-10, 20, 0.0, mv, pc, g)
    else:
        eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 4, 1e-10, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 4, 1e-10, 0.0, mv, pc, g)
    assert not ok and not oko
    assert info["iters"] == 4
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-9, atol=0)
    # the returned vectors are the current Ritz vectors: Rayleigh quotients reproduce eig
    idx = np.arange(1, n + 1, dtype=np.float64)
    a = 1.0 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1.0)
    rq = np.einsum("ij,ij->j", v[:, :n_targ], a @ v[:, :n_targ]) / np.einsum("ij,ij->j", v[:, :n_targ], v[:, :n_targ])
    assert np.allclose(rq, eig[:n_targ], rtol=1e-9)


def test_loose_tolerance_still_takes_two_iterations(ctx, oracle):
    """done(i) can only be set when it > 1 (diaglib.f90:1741): an already converged guess costs two iterations."""
    n, n_targ, n_max = 600, 2, 4
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 50, 1.0e3, 20, 0.0, mv, pc, g)
    eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 50, 1.0e3, 20, 0.0, mv, pc, g)
    assert ok and oko and info["iters"] == 2 and tr.iters == 2
    assert np.allclose(eig[:n_targ], eo[:n_targ], rtol=1e-12)


TORCH_CB_WORKER = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as sla
import torch
torch.cuda.init()                      # torch's HIP runtime first, then the engine (one runtime instance per process)
from diaglib_amd import capi
ctx = capi.Context()
n, n_targ, n_max = 20000, 4, 8
offs = [1, 7, 150]
diags = [np.arange(1, n + 1) * 0.01 + 1.0] + [np.full(n - o, 0.05 / (j + 1)) for j, o in enumerate(offs)]
a = sp.diags(diags, [0] + offs, format="csr")
a = (a + sp.triu(a, 1).T).tocsr()
d = a.diagonal()
want = np.sort(sla.eigsh(a, k=n_targ, which="SA", tol=1e-12)[0])
at = torch.sparse_csr_tensor(torch.from_numpy(a.indptr.astype(np.int64)), torch.from_numpy(a.indices.astype(np.int64)),
                             torch.from_numpy(a.data), size=(n, n), dtype=torch.float64, device="cuda")
dt = torch.from_numpy(d).cuda()
def matvec(x):                         # x: n x m torch view of the device block
    return at @ x.contiguous()
def precnd(fac, x):
    den = dt + fac
    return torch.where(den.abs()[:, None] > 1e-5, x / den[:, None], x)
g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
ev = ctx.panel(g)
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
eig, _, ok, info = ctx.davidson_driver(n, n_targ, n_max, 300, 1e-9, 20, 0.0, matvec, precnd, ev)
assert ok, info
assert np.allclose(eig[:n_targ], want, rtol=1e-9, atol=1e-10), (eig[:n_targ], want)
v = ev.download()[:, :n_targ]
assert np.abs(a @ v - v * eig[:n_targ]).max() < 1e-6
print("OK", info)
"""


def test_torch_sparse_operator_as_device_callback(tmp_path):
    """Device-resident Python callbacks: the blocks arrive as torch tensors aliasing HBM (zero copy), the operator
    is a torch sparse CSR matrix, nothing crosses PCIe inside the solve.  Checked against scipy's eigsh.  Runs in a
    child process that initialises torch's HIP runtime before the engine."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "torch_cb.py"
    script.write_text(TORCH_CB_WORKER.format(root=root))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]


@pytest.mark.parametrize("order,side_stream", [(0, False), (1, True)])
def test_device_callbacks_on_another_stream(ctx, oracle, order, side_stream):
    """the engine's stream is non-blocking, so a device-mode callback that launches on another stream must
    be ordered against it (include/diaglib_amd.h DLA_OPT_CALLBACK_ORDER).  order 0: callback on the legacy null stream
    (torch's default stream) with NO synchronisation of its own; order 1: callback on a private non-blocking stream,
    again without synchronising.  Both must reproduce the host-callback solve."""
    import torch
    n, t, m = 60000, 4, 8
    oracle.synth_setup(n, 0, n)
    w = torch.from_numpy(oracle.synth_w()).cuda()
    d = torch.from_numpy(oracle.synth_diag() - 0.5 * (oracle.synth_w() ** 2).sum(1)).cuda()     # i + 1
    dg = torch.from_numpy(oracle.synth_diag()).cuda()
    # the whole callback (operator AND the copy into the output block) runs on torch's current stream: the default
    # (= legacy null) stream, or a private non-blocking stream made current for the duration of the solve
    side = torch.cuda.Stream() if side_stream else torch.cuda.default_stream()

    def mv(x):
        return d[:, None] * x + 0.5 * (w @ (w.T @ x))

    def pc(fac, x):
        den = dg + fac
        return torch.where(den.abs()[:, None] > 1e-5, x / den[:, None], x)

    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    eo, vo, oko, tr = oracle.davidson(n, t, m, 100, 1e-10, 20, 0.0, oracle.fn("orc_synth_matvec"),
                                      oracle.fn("orc_synth_precnd"), g)
    # torch's lazy initialisation (BLAS handle, workspaces) is not part of the ordering contract
    xx = torch.zeros(n, m, dtype=torch.float64, device="cuda")
    mv(xx); pc(1.0, xx); torch.cuda.synchronize()
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.set_option(capi.OPT_CALLBACK_ORDER, order)
    ctx.sync_python_callbacks = False            # rely on the engine's ordering only, like a compiled caller
    try:
        ev = ctx.panel(g)
        with torch.cuda.stream(side):
            eig, _, ok, info = ctx.davidson_driver(n, t, m, 100, 1e-10, 20, 0.0, mv, pc, ev)
        vec = ev.download()
    finally:
        ctx.sync_python_callbacks = True
        ctx.set_option(capi.OPT_CALLBACK_ORDER, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    assert ok and oko
    assert np.allclose(eig[:t], eo[:t], rtol=1e-11, atol=0)
    assert abs(info["iters"] - tr.iters) <= 1
    _cmp_vecs(vec, vo, t, 1e-6)


def test_wide_gemm_rejects_aliasing(ctx, rng):
    """more than 48 output columns are produced 48 at a time, so the output may not alias the input."""
    n, l = 512, 60
    x = ctx.panel(np.asfortranarray(rng.standard_normal((n, l))))
    c = np.asfortranarray(rng.standard_normal((l, 50)))
    with pytest.raises(capi.DlaError):
        ctx.panel_gemm(x, c, x.col(0, 50))
    z = ctx.panel(n, 50)
    ctx.panel_gemm(x, c, z)
    assert np.abs(z.download() - x.download() @ c).max() < 1e-11


def test_two_host_threads_solve_at_the_same_time(oracle):
    """A16 (the reference's module-level work arrays and timers, diaglib.f90:155-161, make it non-reentrant):
    here the context of the drivers belongs to the calling thread, so two host threads run two different solves --
    different sizes, different operators (sigma), different drivers -- concurrently, each against the oracle."""
    import threading
    specs = [dict(n=300_000, t=8, m=13, sigma=0.5, solver="davidson"), dict(n=200_000, t=4, m=8, sigma=0.25, solver="lobpcg")]
    res, errs = [None, None], []

    def work(i):
        try:
            sp = specs[i]
            c = capi.Context()                       # the CALLING thread's context
            c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
            c.synth_setup(sp["n"], 0, sp["n"], 4, sp["sigma"])
            g = np.zeros((sp["n"], sp["m"]), order="F"); g[np.arange(sp["m"]), np.arange(sp["m"])] = 1.0
            mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
            outs = []
            for _ in range(3):                       # a few solves each, so that the two threads really overlap
                ev = c.panel(g)
                if sp["solver"] == "davidson":
                    eig, _, ok, info = c.davidson_driver(sp["n"], sp["t"], sp["m"], 100, 1e-10, 20, 0.0, mv, pc, ev)
                else:
                    eig, _, ok, info = c.lobpcg_driver(sp["n"], sp["t"], sp["m"], 100, 1e-10, 0.0, mv, pc, ev)
                outs.append((eig.copy(), ok, dict(info)))
            res[i] = (outs, c.h)
        except Exception as exc:                     # noqa: BLE001
            errs.append(repr(exc))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    assert not errs, errs
    assert res[0][1] != res[1][1]                    # two distinct contexts
    for sp, (outs, _) in zip(specs, res):
        n, t, m = sp["n"], sp["t"], sp["m"]
        oracle.lib.orc_synth_setup(n, 0, n, 4, sp["sigma"])
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        if sp["solver"] == "davidson":
            eo, _, oko, tr = oracle.davidson(n, t, m, 100, 1e-10, 20, 0.0, oracle.fn("orc_synth_matvec"), oracle.fn("orc_synth_precnd"), g)
        else:
            eo, _, oko, tr = oracle.lobpcg(n, t, m, 100, 1e-10, 0.0, oracle.fn("orc_synth_matvec"), oracle.fn("orc_synth_precnd"), g)
        for eig, ok, info in outs:
            assert ok and oko
            assert np.allclose(eig[:t], eo[:t], rtol=1e-10, atol=0)
            assert abs(info["iters"] - tr.iters) <= max(1, tr.iters // 10)
            assert np.array_equal(eig, outs[0][0])   # the same thread gets the same bits every time


def test_lobpcg_residual_block_of_tiny_norm_is_not_left_pending_unprojected(ctx):
    """Found by tools/fuzz_multirank.py (seed 78): near convergence LOBPCG's block of preconditioned residuals has norm 1e-8, so
    X^T U measured IN FRONT of the first projection is tiny although the block lies mostly inside span(X); with the pending factor
    (1e8) it is not.  The chain has to test (X^T U) W against the caller's bound, not X^T U: before the fix the closing block came
    back with a Gram matrix that was not positive definite and the driver stopped."""
    n, t, m = 235_047, 29, 36
    try:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ctx.synth_setup(n, 0, n)
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        ev = ctx.panel(g)
        eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 200, 1e-8, 0.0, capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd"), ev)
        assert ok and info["iters"] < 30
        v = ev.download()[:, :t]
        assert np.abs(v.T @ v - np.eye(t)).max() < 1e-12
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)


def test_short_lived_threads_release_their_contexts():
    """The drivers' context belongs to the calling thread and goes away with it: 30 threads that each solve once and end
    leave device memory where it started (each of them holds ~0.25 GB of cached panels while it lives)."""
    import threading
    import torch
    n, t, m = 60_000, 4, 8
    mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
    errs = []

    def work():
        try:
            c = capi.Context()
            c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
            c.synth_setup(n, 0, n)
            g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
            ev = c.panel(g)
            eig, _, ok, _ = c.davidson_driver(n, t, m, 100, 1e-9, 20, 0.0, mv, pc, ev)
            assert ok
            ev.free()
        except Exception as exc:                     # noqa: BLE001
            errs.append(repr(exc))

    def run(k):
        for _ in range(k):
            th = threading.Thread(target=work)
            th.start()
            th.join()

    run(2)                                           # code objects, allocator pools of the runtime
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    run(30)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert not errs, errs
    assert free0 - free1 < 256 << 20, (free0 - free1) / 2 ** 20      # 30 leaked contexts would hold ~7 GB


@pytest.mark.parametrize("solver,n,t,m,max_dav,tol,seed", [("davidson", 2000, 4, 8, 20, 1e-8, 5), ("lobpcg", 2000, 4, 8, 20, 1e-8, 5),
                                                          ("davidson", 1000, 10, 15, 20, 1e-8, 7), ("davidson", 600, 4, 8, 10, 1e-8, 9),
                                                          ("lobpcg", 3000, 16, 21, 20, 1e-8, 11), ("davidson", 3000, 16, 21, 10, 1e-9, 13),
                                                          ("lobpcg", 2500, 32, 37, 20, 1e-8, 15), ("davidson", 2500, 32, 37, 10, 1e-8, 17)])
def test_seeded_random_guess_histories_match_the_oracle(ctx, oracle, solver, n, t, m, max_dav, tol, seed):
    """Round-5 review: with random guesses the tests above allow 10-15 % in the iteration count (one historic LOBPCG case moves
    between 43 and 51 with the last bits of the Gram sums), which would hide a real regression of that size.  On these eight seeded
    uniform guesses -- the reference's dense test matrix, block widths 8 ... 37, restarts at max_dav = 10, both drivers -- the HIP path
    takes EXACTLY the oracle's iteration counts (profiles/r06/iteration_counts.txt: 30 / 30, 30 / 30, 17 / 17, 23 / 23, 35 / 35, 39 / 39,
    23 / 23, 24 / 24; matvec columns equal in seven cases, 236 against 239 in the first); the test holds it to one iteration and 2 %
    of the columns, an order of magnitude tighter than the statistical bound."""
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.asfortranarray(np.random.default_rng(seed).random((n, m)) - 0.5)
    if solver == "davidson":
        e, _, ok, info = ctx.davidson_driver(n, t, m, 400, tol, max_dav, 0.0, mv, pc, g.copy(order="F"))
        eo, _, oko, tr = oracle.davidson(n, t, m, 400, tol, max_dav, 0.0, mv, pc, g)
    else:
        e, _, ok, info = ctx.lobpcg_driver(n, t, m, 400, tol, 0.0, mv, pc, g.copy(order="F"))
        eo, _, oko, tr = oracle.lobpcg(n, t, m, 400, tol, 0.0, mv, pc, g)
    assert ok and oko
    assert abs(info["iters"] - tr.iters) <= 1, (info, tr.iters)
    assert abs(info["matvec_cols"] - tr.matvec_cols) <= 2 + tr.matvec_cols // 50, (info, tr.matvec_cols)
    assert np.allclose(e[:t], eo[:t], rtol=1e-8, atol=0)          # (tol 1e-8 ... 1e-9 in the residual: eigenvalues to its square)
