"""GPU parity of the DEVICE-DRIVEN orthogonalisation chain (hip_engine.hip: ortho_tail_kernel + predicated sweeps)
against the oracle's restatement of ortho_cd (reference diaglib.f90:3185-3341) and ortho_vs_x (:3481-3574), and
against the product's own host-driven loop (same decisions, one host wait per sweep).

The chain takes ortho_vs_x when U is the block that follows X in one panel (the drivers' layout) and every plain
ortho_cd, for k <= 48.  Results are unique up to rounding (Q of the QR factorisation with a positive diagonal), so the
comparison is entry-wise; tolerances: 1e-11 for well-conditioned inputs (rounding amplified by the conditioning of
the block), orthonormality 50 eps."""
import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps
TUNE_CHAIN = 100 + 6      # DLA_OPT_TUNE0 + 6: value 3 = host-driven loop (A/B and parity knob)


def _panel_xu(ctx, x, u):
    big = ctx.panel(np.asfortranarray(np.hstack([x, u])))
    return big, big.col(0, x.shape[1]), big.col(x.shape[1], u.shape[1])


def _syncs(ctx):
    return ctx.stats()["host_syncs"]


@pytest.mark.parametrize("n,m,k,kind", [
    (2000, 13, 13, "random"), (2000, 130, 13, "random"), (4001, 39, 13, "random"),       # odd n: 8-byte path
    (3000, 247, 13, "near_span"),                                                       # U almost inside span(X)
    (3000, 100, 21, "random"), (2500, 74, 37, "random"), (2500, 74, 37, "near_span"),   # 2- and 3-tile blocks
    (1000, 26, 13, "rank_deficient"),                                                   # level-shift ladder
    (1000, 8, 5, "random"), (600, 3, 1, "random"),
])
@pytest.mark.parametrize("schedule", [0, 12, 13, 15, 16])
def test_chain_ortho_vs_x_vs_oracle(ctx, oracle, rng, n, m, k, kind, schedule):
    """(15 / 16: the first factor from U^T U as in the reference / from the projected block's Gram matrix with level shifts: A/B knobs)
    schedule = tune knob 6: 0 the shipped choice (three-pass unless a recent chain needed a level shift), 12 the five-sweep
    schedule, 13 the three-pass one from the first chain on (projection sweeps that measure X^T U of what they store, closing sweep
    without a measurement) -- same result to rounding under each."""
    ctx.set_option(TUNE_CHAIN, schedule)
    x = np.linalg.qr(rng.standard_normal((n, m)))[0]
    if kind == "random":
        u = rng.standard_normal((n, k))
    elif kind == "near_span":
        u = x @ rng.standard_normal((m, k)) + 1e-7 * rng.standard_normal((n, k))
    else:
        u = rng.standard_normal((n, k)); u[:, -1] = u[:, 0] + u[:, 1]; u[:, 2] = 2.0 * u[:, 1]
    x, u = np.asfortranarray(x), np.asfortranarray(u)
    big, px, pu = _panel_xu(ctx, x, u)
    s0 = _syncs(ctx)
    ctx.ortho_vs_x(px, pu)
    used = _syncs(ctx) - s0
    got = pu.download()
    assert used <= 3, used                        # one wait when the expected schedule holds, a few when it does not
    assert np.array_equal(px.download(), x)       # X is read-only
    assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
    assert np.abs(x.T @ got).max() < 50 * EPS
    if kind != "rank_deficient":
        want = oracle.ortho_vs_x(x, u)[0]
        cond = 1e7 if kind == "near_span" else 1.0
        assert np.abs(got - want).max() < 1e-11 * max(1.0, cond * 1e-3), np.abs(got - want).max()
    # same inputs through the host-driven loop of the product: same decisions, results agree to rounding
    big2, px2, pu2 = _panel_xu(ctx, x, u)
    ctx.set_option(TUNE_CHAIN, 3)
    try:
        s0 = _syncs(ctx)
        ctx.ortho_vs_x(px2, pu2)
        used_host = _syncs(ctx) - s0
    finally:
        ctx.set_option(TUNE_CHAIN, 0)
    assert used_host > used
    if kind != "rank_deficient":
        assert np.abs(pu2.download() - got).max() < 1e-12 * (1e4 if kind == "near_span" else 1.0)


@pytest.mark.parametrize("n,k,cond", [(257, 1, 1.0), (257, 5, 1e3), (1000, 13, 1.0), (1000, 13, 1e6), (1000, 13, 1e12),
                                      (2000, 21, 1e5), (2001, 37, 1e8), (1500, 48, 1e3)])
def test_chain_ortho_cd_vs_oracle(ctx, oracle, rng, n, k, cond):
    q = np.linalg.qr(rng.standard_normal((n, k)))[0]
    sv = np.logspace(0, -np.log10(cond), k) if k > 1 else np.ones(1)
    u = np.asfortranarray(q * sv[None, :] @ np.linalg.qr(rng.standard_normal((k, k)))[0])
    p = ctx.panel(u)
    s0 = _syncs(ctx)
    g, ok = ctx.ortho_cd(p)
    assert _syncs(ctx) - s0 <= 3
    got = p.download()
    want, g_want, ok_want, _ = oracle.ortho_cd(u)
    assert ok and ok_want
    assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
    if cond <= 1e6:
        assert np.abs(got - want).max() < 1e-11 * max(1.0, cond * 1e-3)
        # growth = prod ||L^-1||: the first factor comes from a Gram matrix of condition cond^2
        assert g == pytest.approx(g_want, rel=max(1e-8, 100 * cond ** 2 * EPS))
    else:
        assert np.abs(got @ (got.T @ want) - want).max() < 100 * cond * EPS     # same span


def test_chain_gives_up_like_the_host_loop(ctx, oracle, rng):
    """maxit = 1 (test knob): ortho_cd cannot finish on a block with condition 1e6; the chain reports it
    (ok = .false., reference :3252-3254) and ortho_vs_x falls through to the host loop and its QR fallback."""
    n, k = 1200, 13
    q = np.linalg.qr(rng.standard_normal((n, k)))[0]
    u = np.asfortranarray(q * np.logspace(0, -6, k)[None, :])
    ctx.set_option(capi.OPT_ORTHO_MAXIT, 1)
    try:
        p = ctx.panel(u)
        g, ok = ctx.ortho_cd(p)
        assert not ok
        x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, 20)))[0])
        big, px, pu = _panel_xu(ctx, x, u)
        ctx.ortho_vs_x(px, pu)
        got = pu.download()
        assert np.abs(got.T @ got - np.eye(k)).max() < 1e-12
        assert np.abs(x.T @ got).max() < 1e-12
    finally:
        ctx.set_option(capi.OPT_ORTHO_MAXIT, 10)


def test_chain_speculation_recovers_from_a_wrong_plan(ctx, rng):
    """The first call of a shape uses the schedule measured on the reference; an input that needs a different one
    (already orthonormal U: no second macro-iteration, no second projection) must still come out right, and the
    next call of the same kind is then planned from what the device actually did."""
    n, m, k = 3000, 40, 11
    q = np.linalg.qr(rng.standard_normal((n, m + k)))[0]
    x, u = np.asfortranarray(q[:, :m]), np.asfortranarray(q[:, m:])
    for attempt in range(2):
        big, px, pu = _panel_xu(ctx, x, u)
        s0 = _syncs(ctx)
        ctx.ortho_vs_x(px, pu)
        used = _syncs(ctx) - s0
        got = pu.download()
        assert np.abs(got - u).max() < 1e-13
        if attempt == 1:
            assert used == 1, used


@pytest.mark.parametrize("n,m,k", [(300000, 104, 13), (262144, 39, 13), (200000, 74, 37)])
def test_hand_over_forms_give_identical_bits(ctx, rng, n, m, k):
    """the reduction hands its level-2 rows and the reduced matrix to the k x k step with write-through stores, a
    drained store counter and sc1 loads (what gfx942 / gfx950 make of relaxed agent-scope atomics) instead of release / acquire
    fences.  Both forms are in the build (tune knob 5 = 3 selects the fenced one): on reductions with many blocks and several
    output tiles (256 partial blocks -> 8 groups, m / 16 + 1 tiles) a whole ortho_vs_x chain must give the same bits either way,
    repeatedly."""
    x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, m)))[0])
    u = np.asfortranarray(rng.standard_normal((n, k)) + x[:, :k] * 3.0)
    outs = []
    for fenced in (0, 3, 0, 3):
        ctx.set_option(100 + 5, fenced)
        try:
            big = ctx.panel(np.asfortranarray(np.hstack([x, u])))
            ctx.ortho_vs_x(big.col(0, m), big.col(m, k))
            outs.append(big.col(m, k).download())
            big.free()
        finally:
            ctx.set_option(100 + 5, 0)
    q = outs[0]
    assert np.abs(q.T @ q - np.eye(k)).max() < 50 * np.finfo(float).eps and np.abs(x.T @ q).max() < 50 * np.finfo(float).eps
    for o in outs[1:]:
        assert np.array_equal(o, q)
