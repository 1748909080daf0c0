"""dla_expand_project (the drivers' expansion step: ortho_vs_x + operator + projection in one call, reference
diaglib.f90:1790 + 1685 + 1691 for Davidson, :523-529 + 394-397 + 401-403 for LOBPCG) against the three separate entry
points it replaces, on the same inputs.

With device-mode callbacks the call enqueues the operator and the projection sweep BEHIND the orthogonalisation chain and
reads the chain's report at the projection's host wait.  When the chain needs more launches than planned (first call of a
shape, rank-deficient block) what ran behind it is repeated on the finished block: the results must be the same either way,
bit for bit -- the run-ahead changes the order in which the host learns things, not one kernel's arithmetic."""
import ctypes as C

import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _setup(ctx, n):
    ctx.set_shard(n, 0)
    ctx.synth_setup(n, 0, n)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)


def _blocks(rng, n, m, k, kind):
    x = np.linalg.qr(rng.standard_normal((n, m)))[0]
    if kind == "random":
        u = rng.standard_normal((n, k))
    elif kind == "near_span":
        u = x @ rng.standard_normal((m, k)) + 1e-7 * rng.standard_normal((n, k))
    else:
        u = rng.standard_normal((n, k)); u[:, -1] = u[:, 0] + u[:, 1]
    return np.asfortranarray(x), np.asfortranarray(u)


def _apply(ctx, x):
    """A x through the built-in operator (at most 64 columns per call)"""
    px = ctx.panel(x); pax = ctx.panel(np.zeros_like(x))
    for c0 in range(0, x.shape[1], 48):
        w = min(48, x.shape[1] - c0)
        ctx.synth_matvec(px.col(c0, w), pax.col(c0, w))
    return pax.download()


def _run(ctx, mode, x, u, ax, shift, ahead):
    n, m = x.shape
    k = u.shape[1]
    ctx.set_option(capi.OPT_RUN_AHEAD, 1 if ahead else 0)
    basis = ctx.panel(np.asfortranarray(np.hstack([x, u])))
    abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
    s0 = ctx.stats()["host_syncs"]
    h = ctx.expand_project(mode, basis, abasis, m, k, capi.fn_address("dla_synth_matvec"), shift)
    syncs = ctx.stats()["host_syncs"] - s0
    return basis.download(), abasis.download(), h, syncs


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,m,k,kind", [(4000, 26, 13, "random"), (4000, 52, 11, "near_span"), (3000, 26, 13, "rank_deficient"),
                                        (2500, 74, 37, "random"), (3001, 21, 8, "random")])
def test_expand_project_equals_the_three_calls(ctx, rng, mode, n, m, k, kind):
    _setup(ctx, n)
    try:
        x, u = _blocks(rng, n, m, k, kind)
        ax = _apply(ctx, x)
        shift = 0.25 if mode == 1 else 0.0
        # reference result: the separate entry points
        basis = ctx.panel(np.asfortranarray(np.hstack([x, u])))
        abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
        bx, bu = basis.col(0, m), basis.col(m, k)
        ctx.ortho_vs_x(bx, bu)
        ctx.synth_matvec(bu, abasis.col(m, k))
        if shift:
            au = abasis.download(); au[:, m:] += shift * basis.download()[:, m:]
            abasis = ctx.panel(np.asfortranarray(au))
        want_b, want_ab = basis.download(), abasis.download()
        if mode == 0:
            want_h = ctx.gram(basis, abasis.col(m, k))
        else:
            want_h = ctx.gram_lower(basis, abasis)
        # the fused call: first time for this shape in this order (plan from history or the default), then again with the
        # history it has just written, then with the run-ahead switched off
        for ahead in (True, True, False):
            got_b, got_ab, got_h, syncs = _run(ctx, mode, x, u, ax, shift, ahead)
            assert np.array_equal(got_b[:, :m], x)
            assert np.abs(got_b[:, m:] - want_b[:, m:]).max() < 1e-13, kind
            q = got_b[:, m:]
            assert np.abs(q.T @ q - np.eye(k)).max() < 50 * EPS and np.abs(x.T @ q).max() < 50 * EPS
            scale = np.abs(want_ab[:, m:]).max()
            assert np.abs(got_ab[:, m:] - want_ab[:, m:]).max() < 1e-12 * scale
            if mode == 0:
                assert np.abs(got_h - want_h).max() < 1e-11 * np.abs(want_h).max()
            else:
                assert np.abs(np.tril(got_h) - np.tril(want_h)).max() < 1e-11 * np.abs(want_h).max()
        # projection of what is stored: h is consistent with the panels the call left behind
        gb, gab = got_b, got_ab
        ref_h = gb.T @ gab[:, m:] if mode == 0 else np.tril(gb.T @ gab)
        assert np.abs((got_h if mode == 0 else np.tril(got_h)) - ref_h).max() < 1e-10 * np.abs(ref_h).max()
    finally:
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_expand_project_one_wait_when_the_plan_holds(ctx, rng):
    """Steady state (the plan remembered from the previous block of this width fits): ONE host wait for the whole expansion --
    the projection's -- against two for the separate calls; and a block that takes another route (rank deficient after
    well-conditioned ones) costs more waits and a repeated operator call, not a different result."""
    n, m, k = 6000, 39, 13
    _setup(ctx, n)
    try:
        x, u = _blocks(rng, n, m, k, "random")
        ax = _apply(ctx, x)
        _run(ctx, 0, x, u, ax, 0.0, True)                      # writes the history
        _, _, _, syncs = _run(ctx, 0, x, u, ax, 0.0, True)
        assert syncs == 1, syncs
        _, _, _, syncs_off = _run(ctx, 0, x, u, ax, 0.0, False)
        assert syncs_off == 2, syncs_off
        x2, u2 = _blocks(rng, n, m, k, "rank_deficient")
        b_on, ab_on, h_on, syncs_rd = _run(ctx, 0, x2, u2, ax, 0.0, True)
        assert syncs_rd >= 2
        b_off, ab_off, h_off, _ = _run(ctx, 0, x2, u2, ax, 0.0, False)
        assert np.array_equal(b_on, b_off) and np.array_equal(ab_on, ab_off) and np.array_equal(h_on, h_off)
    finally:
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_expand_project_calls_a_user_device_callback_once_unless_it_is_declared_pure(ctx, rng):
    """A caller's own device-mode operator (ordering contract 2: it launches on the engine's stream).  Default: called exactly
    once per block, after the orthogonalisation, like the reference does (diaglib.f90:1685) -- it may count its calls or keep
    state.  With DLA_OPT_RUN_AHEAD = 2 (the caller declares it pure) it runs ahead of the chain's report: once when the chain
    ends as planned, a second time on the same block when it does not.  The statistics never count a dropped call."""
    n, m, k = 3000, 26, 13
    _setup(ctx, n)
    calls, booked = [], {}

    def op(pn, pm, px, pax):
        calls.append(pm[0])
        ctx.lib.dla_synth_matvec(pn, pm, px, pax)

    cb = capi.MATVEC_T(op)
    try:
        ctx.set_option(capi.OPT_CALLBACK_ORDER, 2)
        x, u = _blocks(rng, n, m, k, "random")
        ax = _apply(ctx, x)
        for ahead, kind, expect in ((1, "random", 1), (1, "random", 1), (1, "rank_deficient", 1),
                                    (2, "random", None), (2, "random", 1), (2, "rank_deficient", 2)):
            ctx.set_option(capi.OPT_RUN_AHEAD, ahead)
            x2, u2 = (x, u) if kind == "random" else _blocks(rng, n, m, k, kind)
            basis = ctx.panel(np.asfortranarray(np.hstack([x2, u2])))
            abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
            calls.clear()
            s0 = ctx.stats()
            ctx.expand_project(0, basis, abasis, m, k, C.cast(cb, C.c_void_p).value, 0.0)
            s1 = ctx.stats()
            if expect is not None:
                assert len(calls) == expect and all(c == k for c in calls), (ahead, kind, calls)
            # one operator application is booked however often the device ran it
            booked.setdefault(kind, set()).add((s1["matvec"]["launches"] - s0["matvec"]["launches"],
                                                s1["matvec"]["alg_bytes"] - s0["matvec"]["alg_bytes"]))
            q = basis.download()[:, m:]
            assert np.abs(q.T @ q - np.eye(k)).max() < 50 * EPS
        assert all(len(v) == 1 for v in booked.values()), booked
    finally:
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACK_ORDER, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_expand_project_rejects_bad_sizes(ctx, rng):
    n, m, k = 2000, 13, 13
    _setup(ctx, n)
    try:
        basis = ctx.panel(np.asfortranarray(rng.standard_normal((n, m + k))))
        abasis = ctx.panel(np.zeros((n, m + k), order="F"))
        h = np.zeros((m + k, k), order="F")
        fn = capi.fn_address("dla_synth_matvec")
        call = lambda nn, ldh: ctx.lib.dla_expand_project(ctx.h, 0, nn, m, k, basis.ptr, abasis.ptr, fn, 0.0,
                                                          h.ctypes.data_as(capi.c_dp), ldh)
        assert call(0, m + k) == capi.ERR_ARG and call(-5, m + k) == capi.ERR_ARG
        assert call(n, m + k - 1) == capi.ERR_ARG          # a short leading dimension would be written past
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("n,m,k,kind", [(4000, 26, 13, "random"), (4000, 52, 11, "near_span"), (3000, 42, 21, "random"), (3001, 21, 8, "random")])
def test_expand_project_metric_equals_the_separate_calls(ctx, rng, mode, n, m, k, kind):
    """dla_expand_project_metric (b_ortho_vs_x + bvec + b_ortho + matvec + projection, reference diaglib.f90:2170-2190 and
    :523-529) against the separate entry points on the same inputs, with the library's device operators (A = dla_synth_matvec,
    B = dla_synth_metric): run ahead of the chain's report (first call of a shape, then with its history) and one call after the
    other -- U is B-orthonormal and B-orthogonal to X every time, and the results agree to rounding."""
    _setup(ctx, n)
    mv, bv = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_metric")

    def apply(fn_name, x):
        px = ctx.panel(x); py = ctx.panel(np.zeros_like(x))
        fn = getattr(ctx.lib, fn_name)
        nn, mm = C.c_int(x.shape[0]), C.c_int(0)
        for c0 in range(0, x.shape[1], 48):
            mm.value = min(48, x.shape[1] - c0)
            fn(C.byref(nn), C.byref(mm), C.c_void_p(px.col(c0, mm.value).ptr), C.c_void_p(py.col(c0, mm.value).ptr))
        return py.download()

    try:
        x0, u = _blocks(rng, n, m, k, kind)
        # a B-orthonormal X: X <- X L^-T with L = chol(X^T B X)
        bx0 = apply("dla_synth_metric", x0)
        lx = np.linalg.cholesky(x0.T @ bx0)
        x = np.asfortranarray(x0 @ np.linalg.inv(lx).T)
        bx = apply("dla_synth_metric", x); ax = apply("dla_synth_matvec", x)
        shift = 0.25 if mode == 1 else 0.0
        # the separate entry points
        basis = ctx.panel(np.asfortranarray(np.hstack([x, u])))
        bbasis = ctx.panel(np.asfortranarray(np.hstack([bx, np.zeros((n, k))])))
        abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
        bu_, u_ = bbasis.col(m, k), basis.col(m, k)
        ctx.b_ortho_vs_x(basis.col(0, m), bbasis.col(0, m), u_)
        ctx._chk(ctx.lib.dla_call_matvec(ctx.h, bv, n, k, u_.ptr, bu_.ptr))
        ctx.b_ortho(u_, bu_)
        ctx._chk(ctx.lib.dla_call_matvec(ctx.h, mv, n, k, u_.ptr, abasis.col(m, k).ptr))
        want_b, want_bb, want_ab = basis.download(), bbasis.download(), abasis.download()
        want_ab[:, m:] += shift * want_b[:, m:]
        want_h = want_b.T @ want_ab[:, m:] if mode == 0 else np.tril(want_b.T @ want_ab)
        steady = None
        for ahead in (1, 1, 0):
            ctx.set_option(capi.OPT_RUN_AHEAD, ahead)
            basis = ctx.panel(np.asfortranarray(np.hstack([x, u])))
            bbasis = ctx.panel(np.asfortranarray(np.hstack([bx, np.zeros((n, k))])))
            abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
            s0 = ctx.stats()["host_syncs"]
            h = ctx.expand_project_metric(mode, basis, bbasis, abasis, m, k, mv, bv, shift)
            syncs = ctx.stats()["host_syncs"] - s0
            gb, gbb, gab = basis.download(), bbasis.download(), abasis.download()
            q, bq = gb[:, m:], gbb[:, m:]
            assert np.array_equal(gb[:, :m], x)
            assert np.abs(q.T @ bq - np.eye(k)).max() < 1e-12 and np.abs(bx.T @ q).max() < 1e-12, (kind, ahead)
            assert np.abs(q - want_b[:, m:]).max() < 1e-10 * max(1.0, np.abs(want_b[:, m:]).max())
            assert np.abs(gab[:, m:] - want_ab[:, m:]).max() < 1e-10 * np.abs(want_ab[:, m:]).max()
            got = h if mode == 0 else np.tril(h)
            assert np.abs(got - want_h).max() < 1e-10 * np.abs(want_h).max()
            if ahead and kind == "random":
                steady = syncs          # (second pass: the plan of the first is remembered)
        assert steady in (None, 1), steady       # one host wait for the whole step once the plan holds
    finally:
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("n,m,k,kind", [(4000, 26, 13, "random"), (4000, 52, 11, "near_span"), (2500, 74, 37, "random"),
                                        (6000, 42, 21, "random"), (3001, 21, 8, "random"), (5000, 208, 13, "random")])
@pytest.mark.parametrize("ahead", [True, False])
def test_expand_project_with_the_closing_block_pending(ctx, rng, n, m, k, kind, ahead):
    """mode 3 (LOBPCG's W block): the chain ends where it holds X^T U and U^T U measured on the stored block and the converged
    factor; the closing pass of the reference (diaglib.f90:3543-3544 + one macro-iteration of ortho_cd) comes back as the block
    p = [E ; T] ((m + k) x k).  [X | U_stored] p is the orthonormal block mode 1 delivers; the operator's image follows by
    linearity; the projection comes back already corrected; T is upper triangular with a positive diagonal.
    (Tune knob 6 = 13: the three-pass schedule from the first chain on -- a new context runs the five-sweep one until eight chains
    in a row have needed no level shift.)"""
    try:
        _setup(ctx, n)
        ctx.set_option(100 + 6, 13)
        x, u = _blocks(rng, n, m, k, kind)
        shift = 0.25
        axs = np.asfortranarray(_apply(ctx, x) + shift * x)          # (the X block of A S carries the shift like every block, :397)
        b1, a1, h1, _ = _run(ctx, 1, x, u, axs, shift, ahead)
        b3, a3, h3, _ = _run(ctx, 3, x, u, axs, shift, ahead)
        p = ctx.pending_block(m, k)
        e, t = p[:m], p[m:]
        assert np.array_equal(ctx.pending_factor(k), t)
        assert np.array_equal(np.tril(t, -1), np.zeros_like(t)) and np.all(np.diag(t) > 0)
        if n % 2 == 0:                                    # (an odd row count takes the host-driven schedule: nothing stays pending)
            assert not np.array_equal(t, np.eye(k))       # the device chain did leave its closing block pending
            assert np.abs(e).max() < 1e-3 and np.abs(t - np.eye(k)).max() < 0.5
        w1, w3 = b1[:, m:], x @ e + b3[:, m:] @ t
        assert np.abs(w3.T @ w3 - np.eye(k)).max() < 50 * EPS and np.abs(x.T @ w3).max() < 50 * EPS
        assert np.abs(w3 - w1).max() < 1e-12                                       # the same block (rounding apart)
        assert np.abs(axs @ e + a3[:, m:] @ t - a1[:, m:]).max() < 1e-11 * max(1.0, np.abs(a1).max())
        l1, l3 = np.tril(h1), np.tril(h3)
        assert np.abs(l3 - l1).max() < 1e-11 * max(1.0, np.abs(l1).max())
        assert np.array_equal(b3[:, :m], x)
        # a second fetch returns the same block: it belongs to the call that left it
        assert np.array_equal(ctx.pending_block(m, k), p)
    finally:
        ctx.set_option(100 + 6, 0)
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("n,m,k", [(4000, 26, 13), (6000, 42, 21), (4000, 52, 8), (4000, 208, 13), (5000, 74, 37)])
def test_expand_project_keeps_the_closing_block_of_a_block_that_stays(ctx, rng, n, m, k):
    """mode 4 (a Davidson block): the projection comes back RAW, for the stored block, together with the chain's pending block
    [-S T ; T]; dla_basis_admit completes the closing pass against the caller's basis (here D = I: X is a finished block) and
    turns the raw columns into those of the orthonormal block -- equal to mode 0's.  Both a nearly orthonormal block and a random
    one end pending; either way the pair (stored block, p) describes the block mode 0 stores."""
    try:
        _setup(ctx, n)
        ctx.set_option(100 + 6, 13)
        x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, m)))[0])
        q = rng.standard_normal((n, k)); q -= x @ (x.T @ q); q = np.linalg.qr(q)[0]
        for u in (np.asfortranarray(q @ (np.eye(k) + 1e-10 * rng.standard_normal((k, k)))), np.asfortranarray(rng.standard_normal((n, k)))):
            ax = _apply(ctx, x)
            b0, a0, h0, _ = _run(ctx, 0, x, u, ax, 0.0, True)
            b4, a4, h4, _ = _run(ctx, 4, x, u, ax, 0.0, True)
            p = ctx.pending_block(m, k)
            if k <= 16 and m <= 192:
                assert np.any(p[:m] != 0.0)                 # the three-pass chain ended with its closing pass pending
            l = m + k
            hraw = np.zeros((l, l), order="F"); dmat = np.asfortranarray(np.eye(l)); h = np.zeros((l, l), order="F")
            hraw[:m, :m] = x.T @ ax
            h[:, m:] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied)
            e, t = p[:m], p[m:]
            assert np.array_equal(np.tril(t, -1), np.zeros_like(t)) and np.all(np.diag(t) > 0)
            assert np.array_equal(dmat[:, m:], p) and np.array_equal(dmat[:m, :m], np.eye(m))
            w4 = x @ e + b4[:, m:] @ t
            assert np.abs(w4.T @ w4 - np.eye(k)).max() < 50 * EPS and np.abs(x.T @ w4).max() < 50 * EPS
            assert np.abs(w4 - b0[:, m:]).max() < 1e-12
            assert np.abs(h[:, m:] - h0).max() < 1e-11 * max(1.0, np.abs(h0).max())
            assert np.abs(ax @ e + a4[:, m:] @ t - a0[:, m:]).max() < 1e-11 * max(1.0, np.abs(a0).max())
            # coefficients for the stored columns: D c
            cfull = np.asfortranarray(rng.standard_normal((l, 3)))
            want = dmat @ cfull
            ctx.basis_fold(l, dmat, cfull)
            assert np.abs(cfull - want).max() < 1e-13
    finally:
        ctx.set_option(100 + 6, 0)
        ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)
