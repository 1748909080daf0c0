"""A basis that grows block by block through dla_expand_project mode 4 (the Davidson drivers' expansion step, reference
diaglib.f90:1790 + 1685 + 1691) with the blocks' closing passes left pending: the panel holds what the device chains stored, the
caller's upper-triangular D (dla_basis_admit) holds what they left undone.  Whatever schedule a chain took -- three-pass, five-sweep,
sweep-per-update beyond 192 columns, level shifts -- and whatever it left pending,

    (panel D)^T (panel D) = I      and      h = (panel D)^T A (panel D)

to rounding: the property the round-4 review asked for (the stored basis alone need not be orthonormal)."""
import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
EPS = np.finfo(np.float64).eps


def _grow(ctx, rng, n, k, nb, kind, pending, mode=4):
    mv = capi.fn_address("dla_synth_matvec")
    ld = nb * k
    x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, k)))[0])
    basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - k))])))
    abasis = ctx.panel(np.zeros((n, ld), order="F"))
    ctx.synth_matvec(basis.col(0, k), abasis.col(0, k))
    hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
    b = basis.download(); ab = abasis.download()
    hraw[:k, :k] = b[:, :k].T @ ab[:, :k]; h[:k, :k] = hraw[:k, :k]
    n_pending = 0
    if mode == 5:                     # the device keeps D as well: every block is announced, identity ones included
        ctx.basis_sync(0, 0)
        ctx.basis_sync(0, k, dmat)
    for blk in range(1, nb):
        m = blk * k
        if kind == "inside":          # mostly inside span(X): the projection removes almost everything
            u = b[:, :m] @ rng.standard_normal((m, k)) + 1e-6 * rng.standard_normal((n, k))
        elif kind == "dependent":     # numerically rank deficient after the projection: level shifts
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            u[:, -1] = u[:, 0] * (1.0 + 1e-13) + 1e-14 * rng.standard_normal(n)
        else:
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
        basis.col(m, k).upload(np.asfortranarray(u))
        h4 = ctx.expand_project(mode if pending else 0, basis, abasis, m, k, mv, 0.0)
        p = ctx.pending_block(m, k) if pending else np.asfortranarray(np.vstack([np.zeros((m, k)), np.eye(k)]))
        applied = ctx.pending_applied if pending else False
        n_pending += int(np.any(p[:m] != 0.0) or not np.array_equal(p[m:], np.eye(k)))
        h[:m + k, m:m + k] = h4
        ctx.basis_admit(m, k, p, hraw, dmat, h, applied=applied)
        if mode == 5:
            ctx.basis_sync(m, k, dmat)
        b = basis.download()
    ab = abasis.download()
    return b, ab, dmat, h, n_pending


@pytest.mark.parametrize("knob", [0, 12, 13])
@pytest.mark.parametrize("k,nb,kind", [(1, 8, "random"), (3, 8, "random"), (13, 6, "random"), (13, 18, "random"), (13, 6, "inside"),
                                       (13, 6, "dependent"), (8, 10, "dependent"), (21, 5, "random"), (37, 4, "random")])
def test_basis_with_pending_blocks_is_orthonormal_and_projects_exactly(ctx, rng, knob, k, nb, kind):
    n = 6000
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ctx.set_option(100 + 6, knob)
        b, ab, dmat, h, n_pending = _grow(ctx, rng, n, k, nb, kind, True)
        l = nb * k
        v = b @ dmat
        assert np.abs(v.T @ v - np.eye(l)).max() < 50 * EPS, (np.abs(v.T @ v - np.eye(l)).max(), n_pending)
        href = v.T @ (ab @ dmat)
        assert np.abs(np.triu(h - href)).max() < 1e-13 * np.abs(href).max()
        if knob == 13 and k <= 16 and kind == "random":
            assert n_pending >= min(nb - 1, 192 // k - 1) // 2      # the three-pass chains do leave their closing passes pending
    finally:
        ctx.set_option(100 + 6, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("k,nb,kind", [(1, 8, "random"), (3, 8, "random"), (13, 6, "random"), (13, 22, "random"), (16, 20, "random"), (13, 24, "dependent"), (13, 6, "inside"),
                                       (13, 6, "dependent"), (8, 10, "dependent"), (13, 20, "dependent"), (13, 20, "inside")])
def test_basis_kept_on_the_device_projects_exactly(ctx, rng, k, nb, kind):
    """mode 5: the chains project with X (D D^T) X^T, so the stored columns may be as far from orthonormal as the host algebra
    tolerates (max |S| < 0.05) -- the finished basis panel D is orthonormal to rounding all the same, over every basis width the device
    copy takes (320 columns: the last blocks run the sweep-per-update schedule beyond 192)."""
    n = 6000
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        b, ab, dmat, h, n_pending = _grow(ctx, rng, n, k, nb, kind, True, mode=5)
        l = nb * k
        v = b @ dmat
        assert np.abs(v.T @ v - np.eye(l)).max() < 50 * EPS, (np.abs(v.T @ v - np.eye(l)).max(), n_pending)
        href = v.T @ (ab @ dmat)
        assert np.abs(np.triu(h - href)).max() < 1e-13 * np.abs(href).max()
        if kind == "random":
            assert n_pending >= (nb - 1) // 2
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_a_chain_that_stops_half_way_is_continued_on_the_host_with_d(ctx, rng):
    """mode 5 with ortho_cd's iteration limit at 2: the device chain stops (status 2: ortho_cd out of iterations), the reference's
    Householder fallback runs (:3534 / :3549) and the host-driven loop goes on -- projecting with X (D D^T) X^T like the device
    (BlockOps::basis_dd), because the stored columns are not orthonormal.  (Found by tests/test_spmm_sharded.py: a banded operator's
    blocks do run into ortho_cd's limit.)"""
    n, k, nb = 5000, 8, 7
    mv = capi.fn_address("dla_synth_matvec")
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ld = nb * k
        x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, k)))[0])
        basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - k))])))
        abasis = ctx.panel(np.zeros((n, ld), order="F"))
        ctx.synth_matvec(basis.col(0, k), abasis.col(0, k))
        hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
        b = basis.download(); ab = abasis.download()
        hraw[:k, :k] = b[:, :k].T @ ab[:, :k]; h[:k, :k] = hraw[:k, :k]
        ctx.basis_sync(0, 0); ctx.basis_sync(0, k, dmat)
        stopped = 0
        for blk in range(1, nb):
            m = blk * k
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            if blk >= 4:                                   # (the first blocks leave pending parts behind: D is not the identity)
                u[:, 1:] = u[:, :1] + 1e-9 * rng.standard_normal((n, k - 1))       # needs level shifts and several macro-iterations
                ctx.set_option(capi.OPT_ORTHO_MAXIT, 2)
            basis.col(m, k).upload(np.asfortranarray(u))
            syncs = ctx.stats()["host_syncs"]
            h4 = ctx.expand_project(5, basis, abasis, m, k, mv, 0.0)
            stopped += int(blk >= 4 and ctx.stats()["host_syncs"] - syncs > 10)      # (the host-driven loop waits per operation)
            ctx.set_option(capi.OPT_ORTHO_MAXIT, 10)
            p = ctx.pending_block(m, k)
            h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied)
            ctx.basis_sync(m, k, dmat)
            b = basis.download()
        assert not np.array_equal(dmat, np.eye(ld)) and stopped >= 1
        v = b @ dmat
        assert np.abs(v.T @ v - np.eye(ld)).max() < 200 * EPS, np.abs(v.T @ v - np.eye(ld)).max()
        href = v.T @ (abasis.download() @ dmat)
        assert np.abs(np.triu(h - href)).max() < 1e-12 * np.abs(href).max()
    finally:
        ctx.set_option(capi.OPT_ORTHO_MAXIT, 10)
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_full_size_basis_of_twenty_blocks(ctx):
    """The round-4 review's property test at the benchmark's size: n = 2e6, 13-column blocks, the basis filled to 20 blocks (260
    columns) through dla_expand_project mode 5 with the caller's own D -- random blocks, blocks with most of their norm inside
    span(X), blocks inside span(X) to 1e-7.  Checked through dla_gram on the device panels:  D^T (X_c^T X_c) D = I  to 50 eps and
    h = D^T (X_c^T A X_c) D."""
    n, k, nb = 2_000_000, 13, 20
    mv = capi.fn_address("dla_synth_matvec")
    rng = np.random.default_rng(11)
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ld = nb * k
        basis = ctx.panel(n, ld); abasis = ctx.panel(n, ld)
        first = basis.col(0, k)
        ctx.fill_guess(first, 3, 0)
        ctx.check_guess(first)                                   # orthonormal first block
        x0 = first.download()
        ctx.synth_matvec(first, abasis.col(0, k))
        hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
        hraw[:k, :k] = ctx.gram(first, abasis.col(0, k)); h[:k, :k] = hraw[:k, :k]
        ctx.basis_sync(0, 0); ctx.basis_sync(0, k, dmat)
        n_pending = 0
        for blk in range(1, nb):
            m = blk * k
            u = rng.standard_normal((n, k)) * (1e-7 if blk % 5 == 4 else 0.3 / np.sqrt(n) * 30.0) + x0 @ rng.standard_normal((k, k))
            basis.col(m, k).upload(np.asfortranarray(u))
            h4 = ctx.expand_project(5, basis, abasis, m, k, mv, 0.0)
            p = ctx.pending_block(m, k)
            n_pending += int(np.any(p[:m] != 0.0) or not np.array_equal(p[m:], np.eye(k)))
            h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied)
            ctx.basis_sync(m, k, dmat)
        g = ctx.gram(basis, basis)
        e = np.abs(dmat.T @ g @ dmat - np.eye(ld)).max()
        assert e < 50 * EPS, (e, n_pending)
        assert n_pending >= nb // 2
        ga = ctx.gram(basis, abasis)
        href = dmat.T @ ga @ dmat
        assert np.abs(np.triu(h - href)).max() < 1e-12 * np.abs(href).max()
    finally:
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_driver_with_restarts_returns_orthonormal_vectors_at_full_size(ctx):
    """... and the driver itself on the benchmark's random-guess leg (42 iterations, two restarts: D is folded into the restart's
    coefficients and reset): the Ritz vectors it returns are orthonormal to 1e-13 and the residuals below the tolerance."""
    n, t, m = 2_000_000, 8, 13
    try:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ctx.synth_setup(n, 0, n)
        ev = ctx.panel(n, m)
        ctx.fill_guess(ev, 2, 2000)
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 100, 2e-13, 20, 0.0, capi.fn_address("dla_synth_matvec"),
                                               capi.fn_address("dla_synth_precnd"), ev)
        assert ok and info["restarts"] >= 1
        g = ctx.gram(ev, ev)
        assert np.abs(g - np.eye(m))[:t, :t].max() < 1e-13
        ax = ctx.panel(n, m)
        ctx.synth_matvec(ev, ax)
        x_h, ax_h = ev.download(), ax.download()
        res = np.sqrt(((ax_h - x_h * eig[None, :]) ** 2).sum(0))[:t] / np.abs(eig[:t])
        assert res.max() < 1e-10, res
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)


def test_mode_5_refuses_a_basis_the_device_copy_does_not_describe(ctx, rng):
    n, k = 4000, 8
    mv = capi.fn_address("dla_synth_matvec")
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, 2 * k)))[0])
        basis = ctx.panel(np.asfortranarray(np.hstack([x0, rng.standard_normal((n, k))])))
        abasis = ctx.panel(np.zeros((n, 3 * k), order="F"))
        ctx.basis_sync(0, 0)
        ctx.basis_sync(0, k, np.asfortranarray(np.eye(3 * k)))          # one block announced, two stored
        with pytest.raises(capi.DlaError):
            ctx.expand_project(5, basis, abasis, 2 * k, k, mv, 0.0)
        with pytest.raises(capi.DlaError):                                # blocks arrive in order
            ctx.basis_sync(2 * k, k, np.asfortranarray(np.eye(3 * k)))
        ctx.basis_sync(k, k, np.asfortranarray(np.eye(3 * k)))
        ctx.expand_project(5, basis, abasis, 2 * k, k, mv, 0.0)
        p = ctx.pending_block(2 * k, k)
        b = basis.download()
        v = b[:, :2 * k] @ p[:2 * k] + b[:, 2 * k:] @ p[2 * k:]
        # (identity D: the closing block the chain hands over is completed by dla_basis_admit; here only its T part is checked)
        assert np.abs(np.triu(p[2 * k:], 0) - p[2 * k:]).max() == 0.0 and np.isfinite(v).all()
    finally:
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_pending_blocks_can_be_switched_off_per_context(ctx, rng):
    """DLA_OPT_PENDING_BLOCKS = 0: modes 3 / 4 finish every block in memory (nothing comes back pending); same basis either way."""
    n, k, nb = 6000, 13, 5
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ctx.set_option(100 + 6, 13)
        outs = []
        for on in (1, 0):
            ctx.set_option(capi.OPT_PENDING_BLOCKS, on)
            assert ctx.get_option(capi.OPT_PENDING_BLOCKS) == on
            b, ab, dmat, h, n_pending = _grow(ctx, np.random.default_rng(7), n, k, nb, "random", True)
            assert (n_pending > 0) == bool(on)
            outs.append(b @ dmat)
        assert np.abs(outs[0] - outs[1]).max() < 1e-12
    finally:
        ctx.set_option(capi.OPT_PENDING_BLOCKS, 1)
        ctx.set_option(100 + 6, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


@pytest.mark.parametrize("how", ["host_loop_knob", "lds_limit_down", "wide_block", "mode6"])
def test_a_call_that_cannot_be_exact_still_projects_against_the_finished_basis(ctx, rng, how):
    """Round-5 advisor: a basis grown through mode 5 holds pending blocks (stored columns orthonormal to 0.05 only); when a later
    call cannot use the device's exact projection -- the host-loop knob, an engine whose LDS limit went down between two calls of a
    solve (basis_exact_ok() false), a block wider than 16 columns, or the caller's own request (mode 6) -- the block must still come
    out orthogonal to the FINISHED basis panel*D, not to the stored columns: the host-driven loop multiplies X^T U by D D^T."""
    n, k, nb = 6000, 13, 6
    mv = capi.fn_address("dla_synth_matvec")
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        b, ab, dmat, h, n_pending = _grow(ctx, rng, n, k, nb, "random", True, mode=5)
        assert n_pending >= 2 and not np.array_equal(dmat, np.eye(nb * k))
        m = nb * k
        kk = 17 if how == "wide_block" else k
        v = b @ dmat
        u = v @ rng.standard_normal((m, kk)) + 0.3 * rng.standard_normal((n, kk))
        basis = ctx.panel(np.asfortranarray(np.hstack([b, u])))
        abasis = ctx.panel(np.asfortranarray(np.hstack([ab, np.zeros((n, kk))])))
        if how == "host_loop_knob":
            ctx.set_option(100 + 6, 3)
        if how == "lds_limit_down":
            ctx.set_option(100 + 6, 14)                                   # (basis_exact_ok() answers false, like an engine under the 64 KiB limit)
        h4 = ctx.expand_project(6 if how == "mode6" else 5, basis, abasis, m, kk, mv, 0.0)
        p = ctx.pending_block(m, kk)
        assert np.array_equal(p, np.vstack([np.zeros((m, kk)), np.eye(kk)]))          # finished in memory
        unew = basis.download()[:, m:]
        assert np.abs(v.T @ unew).max() < 50 * EPS, np.abs(v.T @ unew).max()
        assert np.abs(unew.T @ unew - np.eye(kk)).max() < 50 * EPS
        # ... and NOT merely against the stored columns (the defect: |X_c^T X_c - I| of what was removed stayed in the block)
        aun = abasis.download()[:, m:]
        href = np.vstack([b.T @ aun, unew.T @ aun])
        assert np.abs(h4 - href).max() < 1e-12 * np.abs(href).max()
    finally:
        ctx.set_option(100 + 6, 0)
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_a_basis_with_pending_blocks_goes_on_beyond_the_device_copy(ctx, rng):
    """The device copy of D holds 320 columns; the engine's host copy has no limit.  A basis with pending blocks that grows beyond
    320 columns is neither refused nor -- the defect the round-5 advisor found -- projected against its unfinished stored columns: the
    blocks past the copy are finished in memory by the host-driven loop with X^T U multiplied by D D^T.  22 blocks of 16 columns
    (352): the first 19 through the device chains with pending parts, the last ones on the host."""
    n, k, nb = 5000, 16, 22
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        b, ab, dmat, h, n_pending = _grow(ctx, rng, n, k, nb, "random", True, mode=5)
        assert n_pending >= 8 and np.array_equal(dmat[320:, 320:], np.eye(nb * k - 320))      # (nothing pending past the copy)
        l = nb * k
        v = b @ dmat
        assert np.abs(v.T @ v - np.eye(l)).max() < 50 * EPS, (np.abs(v.T @ v - np.eye(l)).max(), n_pending)
        href = v.T @ (ab @ dmat)
        assert np.abs(np.triu(h - href)).max() < 1e-13 * np.abs(href).max()
    finally:
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


def test_blocks_of_one_and_two_column_tiles_in_one_basis(ctx, rng):
    """What davidson_core produces at n_max = 21 (BASELINE cfg 4): blocks of 21 ... 17 columns while fewer than five roots are
    locked -- finished in memory, D = I there -- then blocks of 16 and fewer columns that keep their closing passes pending, and (a C
    caller may do that) a wide block behind them, which the host-driven loop finishes against panel*D."""
    n = 6000
    widths = [21, 21, 20, 18, 16, 12, 8, 8, 19]
    mv = capi.fn_address("dla_synth_matvec")
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ld = sum(widths)
        x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, widths[0])))[0])
        basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - widths[0]))])))
        abasis = ctx.panel(np.zeros((n, ld), order="F"))
        k0 = widths[0]
        ctx.synth_matvec(basis.col(0, k0), abasis.col(0, k0))
        hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
        b = basis.download(); ab = abasis.download()
        hraw[:k0, :k0] = b[:, :k0].T @ ab[:, :k0]; h[:k0, :k0] = hraw[:k0, :k0]
        ctx.basis_sync(0, 0); ctx.basis_sync(0, k0, dmat)
        m, pending = k0, []
        for k in widths[1:]:
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            basis.col(m, k).upload(np.asfortranarray(u))
            h4 = ctx.expand_project(5, basis, abasis, m, k, mv, 0.0)
            p = ctx.pending_block(m, k)
            pending.append(bool(np.any(p[:m] != 0.0) or not np.array_equal(p[m:], np.eye(k))))
            h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied)
            ctx.basis_sync(m, k, dmat)
            b = basis.download(); m += k
        assert not any(pending[:3]) and any(pending[3:7]) and not pending[7]      # wide: finished; narrow: pending; wide again: finished
        v = b @ dmat
        assert np.abs(v.T @ v - np.eye(ld)).max() < 50 * EPS, np.abs(v.T @ v - np.eye(ld)).max()
        href = v.T @ (abasis.download() @ dmat)
        assert np.abs(np.triu(h - href)).max() < 1e-13 * np.abs(href).max()
    finally:
        ctx.basis_sync(0, 0)
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_shard(-1, 0)


RECOVERY_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np
from diaglib_amd import capi
from oracle.pyoracle import Oracle
ctx, o = capi.Context(), Oracle()
n, t, m = 60000, 8, 13
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n); o.synth_setup(n, 0, n)
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
solver = sys.argv[1]
if solver == "davidson":
    e, v, ok, info = ctx.davidson_driver(n, t, m, 200, 1e-10, 20, 0.0, mv, pc, g.copy(order="F"))
    eo, vo, oko, tr = o.davidson(n, t, m, 200, 1e-10, 20, 0.0, o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd"), g)
else:
    e, v, ok, info = ctx.lobpcg_driver(n, t, m, 200, 1e-10, 0.0, mv, pc, g.copy(order="F"))
    eo, vo, oko, tr = o.lobpcg(n, t, m, 200, 1e-10, 0.0, o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd"), g)
assert ok and oko
assert np.allclose(e[:t], eo[:t], rtol=1e-11, atol=0), (e[:t], eo[:t])
assert abs(info["iters"] - tr.iters) <= 1, (info, tr.iters)
assert np.abs(v[:, :t].T @ v[:, :t] - np.eye(t)).max() < 1e-12
print("RECOVERED", info["iters"], tr.iters, "waits", ctx.stats()["host_syncs"], flush=True)
"""


@pytest.mark.parametrize("solver,nth", [("davidson", 2), ("davidson", 4), ("lobpcg", 3)])
def test_a_closing_pass_that_fails_is_recovered_by_finishing_the_block_in_memory(tmp_path, solver, nth):
    """Round-5 advisor (low): the pending criterion bounds S, nothing bounds |D| |T|, so the closing factor I - F^T F of a pending block
    may fail to factor -- before r06 the Fortran driver answered that with `error stop`.  The failure is forced here
    ($DIAGLIB_AMD_FAIL_CLOSE = n: the n-th closing pass reports it; a child process, because the counter belongs to the thread):
    the Davidson driver finishes the block in memory through dla_expand_project mode 6 (host-driven loop against panel*D), LOBPCG's
    mode 3 repeats the expansion without anything pending -- same eigenvalues and iteration count as the oracle either way."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(RECOVERY_WORKER.format(root=root))
    waits = []
    for env_n in (None, str(nth)):
        env = dict(os.environ)
        env.pop("DIAGLIB_AMD_FAIL_CLOSE", None)
        if env_n:
            env["DIAGLIB_AMD_FAIL_CLOSE"] = env_n
        p = subprocess.run([sys.executable, str(script), solver], capture_output=True, text=True, timeout=600, env=env)
        assert p.returncode == 0 and "RECOVERED" in p.stdout, (env_n, p.stdout[-1500:], p.stderr[-3000:])
        waits.append(int(p.stdout.split("waits")[1].split()[0]))
    assert waits[1] > waits[0], waits          # (the recovery did run: the repeated orthogonalisation waits for the host)
