"""GPU parity against the committed golden fixtures (outputs of the UNMODIFIED reference, see
tests/golden/make_golden.py) and, at BASELINE.json's full sizes, through size-independent
properties (orthonormality, linearity, Ritz identities, residual of the returned eigenpairs)."""
import json
import os

import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
EPS = np.finfo(np.float64).eps


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "reference_fixtures.npz"), allow_pickle=False)


def test_ortho_cd_vs_reference(ctx, gold):
    for i in range(int(gold["ocd_count"])):
        u, want, g_want = gold[f"ocd{i}_in"], gold[f"ocd{i}_out"], float(gold[f"ocd{i}_growth"])
        p = ctx.panel(u)
        g, ok = ctx.ortho_cd(p)
        got = p.download()
        k = u.shape[1]
        sv = np.linalg.svd(u, compute_uv=False)
        cond = sv[0] / max(sv[-1], 1e-300)
        assert ok == bool(gold[f"ocd{i}_ok"])
        if cond < 1e4:
            assert np.abs(got - want).max() < 1e-12 and g == pytest.approx(g_want, rel=1e-9)
        elif cond < 1e14:
            assert np.abs(got.T @ got - np.eye(k)).max() < 50 * EPS
            assert np.abs(got @ (got.T @ want) - want).max() < 100 * cond * EPS
        else:
            assert np.all(np.isfinite(got))


def test_ortho_vs_x_b_ortho_helpers_vs_reference(ctx, gold):
    for i in range(int(gold["ovx_count"])):
        x, u, want = gold[f"ovx{i}_x"], gold[f"ovx{i}_u"], gold[f"ovx{i}_out"]
        pu = ctx.panel(u)
        ctx.ortho_vs_x(ctx.panel(x), pu)
        assert np.abs(pu.download() - want).max() < 1e-11
    pu = ctx.panel(gold["bo_u"])
    ctx.b_ortho_vs_x(ctx.panel(gold["bo_x"]), ctx.panel(gold["bo_bx"]), pu)
    assert np.abs(pu.download() - gold["bo_vsx_out"]).max() < 1e-11
    p1, p2 = ctx.panel(gold["bo_vsx_out"]), ctx.panel(gold["bo_bu"])
    ctx.b_ortho(p1, p2)
    assert np.abs(p1.download() - gold["bo_u_out"]).max() < 1e-12 and np.abs(p2.download() - gold["bo_bu_out"]).max() < 1e-12
    p = ctx.panel(gold["cg_in"]); ctx.check_guess(p)
    assert np.abs(p.download() - gold["cg_out"]).max() < 1e-12
    ux, up = ctx.get_coeffs(gold["gc_a_red"], int(gold["gc_len_u"]), int(gold["gc_n_max"]), int(gold["gc_n_act"]))
    assert np.array_equal(ux, gold["gc_ux"]) and np.abs(up - gold["gc_up"]).max() < 1e-12
    assert capi.norm_est(gold["ne_in"]) == pytest.approx(float(gold["ne_out"]), rel=1e-15)


def _guess(kind, n, m, seed):
    if kind == "unit":
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        return g
    return np.asfortranarray(np.random.default_rng(seed).random((n, m)) - 0.5)


@pytest.mark.parametrize("name", ["dav_n1000_unit", "dav_n2000_unit", "dav_n2000_rand", "dav_n600_rand_dav10",
                                  "lob_n1000_unit", "lob_n2000_unit", "lob_n2000_rand", "lob_n800_shift",
                                  "gdav_n600_unit", "gdav_n600_rand", "glob_n600_unit"])
def test_drivers_vs_reference_results(ctx, oracle, gold, name):
    """Reference's dense test matrix (main.f90:311-317), host callbacks: eigenvalues, eigenvectors and
    iteration counts against what the unmodified reference produced for the same guess."""
    sp = next(s for s in json.loads(str(gold["driver_specs"])) if s["name"] == name)
    n, t, m = sp["n"], sp["n_targ"], sp["n_max"]
    g = _guess(sp["guess"], n, m, sp["seed"])
    oracle.dense_setup(n)                      # the C operator is only the callback here
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if sp.get("gen"):
        oracle.metric_setup(n); bv = oracle.fn("orc_metric_matvec")      # generalised problem: B through a host callback
    if sp["solver"] == "gen_davidson":
        eig, vec, ok, info = ctx.gen_david_driver(n, t, m, sp["max_iter"], sp["tol"], sp["max_dav"], sp["shift"], mv, pc, bv, g)
    elif sp["solver"] == "lobpcg" and sp.get("gen"):
        eig, vec, ok, info = ctx.lobpcg_driver(n, t, m, sp["max_iter"], sp["tol"], sp["shift"], mv, pc, g, bvec=bv)
    elif sp["solver"] == "davidson":
        eig, vec, ok, info = ctx.davidson_driver(n, t, m, sp["max_iter"], sp["tol"], sp["max_dav"], sp["shift"], mv, pc, g)
    else:
        eig, vec, ok, info = ctx.lobpcg_driver(n, t, m, sp["max_iter"], sp["tol"], sp["shift"], mv, pc, g)
    assert ok and bool(gold[name + "_ok"])
    if name == "gdav_n600_rand":
        # This run restarts once (iteration 20).  The UNMODIFIED reference zeroes bspace at the restart
        # (diaglib.f90:2196-2200) and then "converges" with ok=.true. to eigenvalues ~1e-15 --
        # the fixture records that.  Our drivers keep the kept block's B*x and must
        # return the true generalised eigenvalues instead.
        import scipy.linalg as sla
        assert np.abs(gold[name + "_eig"][:t]).max() < 1e-10
        idx = np.arange(1, n + 1.0); a = 1.0 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1.0)
        smat = oracle.metric_setup(n)
        assert np.allclose(eig[:t], sla.eigh(a, smat, eigvals_only=True)[:t], rtol=0, atol=1e-7)
        return
    assert np.allclose(eig[:t], gold[name + "_eig"][:t], rtol=1e-10, atol=0)
    it_ref = int(gold[name + "_tr_iters"])
    if sp["guess"] == "unit":
        assert info["iters"] == it_ref
    else:
        assert abs(info["iters"] - it_ref) <= max(1, it_ref // 10), (info, it_ref)
    if sp["solver"] in ("davidson", "gen_davidson") and sp["guess"] == "unit":
        assert info["restarts"] == int(gold[name + "_tr_restarts"])
    if name + "_evec" in gold.files:
        ev = vec[:, :t]; ev = ev * np.sign(ev[np.abs(ev).argmax(0), np.arange(t)])
        assert np.abs(ev - gold[name + "_evec"]).max() < 1e-5


# ------------------------------------------------------------------ full BASELINE sizes: properties
def _rand_panel(ctx, n, m):
    p = ctx.panel(n, m)
    ctx.random_fill(p)            # device-side generator: no 200 MB host arrays
    return p


@pytest.mark.parametrize("n,l,k", [(500_000, 260, 13), (2_000_000, 39, 13),
                                   (2_000_000, 420, 21),      # BASELINE cfg 4: lda = 20 blocks of n_max = 21
                                   (10_000_000, 111, 37)])    # BASELINE cfg 5: [X | P | W] of n_max = 37
def test_full_size_block_algebra_properties(ctx, n, l, k):
    x, u1, u2 = _rand_panel(ctx, n, l), _rand_panel(ctx, n, k), ctx.panel(n, k)
    # u2 = u1 + 0.5 * (X C): linearity of the Gram kernel in U
    rng = np.random.default_rng(7)
    c = np.asfortranarray(rng.standard_normal((l, k)) * 1e-2)
    ctx.lib.dla_copy(ctx.h, u2.ptr, u1.ptr, 8 * n * k)
    ctx.panel_update(x, -c, u2)                       # u2 = u1 + X c
    g1, g2, gx = ctx.gram(x, u1), ctx.gram(x, u2), ctx.gram(x, x)
    assert np.allclose(g2, g1 + gx @ c, rtol=1e-11, atol=1e-9 * np.abs(gx).max())
    assert np.abs(gx - gx.T).max() <= 1e-12 * np.abs(gx).max()
    # ortho_vs_x on a trusted orthonormal X: result orthonormal and orthogonal to X
    ctx.ortho_cd(x)
    ctx.ortho_vs_x(x, u2)
    assert np.abs(ctx.gram(u2, u2) - np.eye(k)).max() < 50 * EPS
    assert np.abs(ctx.gram(x, u2)).max() < 50 * EPS
    # ortho_cd is idempotent on an orthonormal block: growth ~ 1, nothing moves beyond rounding
    before = ctx.gram(x, u2)
    g, ok = ctx.ortho_cd(u2)
    assert ok and abs(g - 1.0) < 1e-10
    assert np.abs(ctx.gram(x, u2) - before).max() < 50 * EPS


def _unit_guess_panel(ctx, n, m):
    ev = ctx.panel(n, m).zero()
    top = np.zeros((m, m), order="F"); np.fill_diagonal(top, 1.0)
    # e_1..e_m without a host array of n x m: zero on the device, then the leading m x m block column by column
    for j in range(m):
        ctx._chk(ctx.lib.dla_upload(ctx.h, ev.ptr + 8 * n * j, top[:, j].ctypes.data, 8 * m))
    return ev


@pytest.mark.parametrize("solver,n,t,m,max_dav,guess", [
    ("davidson", 500_000, 8, 13, 20, "unit"),        # BASELINE cfg 2
    ("lobpcg", 2_000_000, 8, 13, 20, "unit"),        # cfg 3
    ("davidson", 2_000_000, 16, 21, 20, "unit"),     # cfg 4 (one GPU holds what the 8-GPU config shards: 2 x 6.7 GB panels)
    ("lobpcg", 10_000_000, 32, 37, 20, "unit"),      # cfg 5 (3 basis panels of 8.9 GB, kept twice)
    ("davidson", 10_000_000, 32, 37, 10, "seed2"),   # cfg 5 "restart / thick-restart stress" (SURVEY 8d): max_dav = 10,
])                                                   # seed-2 random guess on the leading 6000 rows (dla_fill_guess)
def test_full_size_solve_residual(ctx, solver, n, t, m, max_dav, guess):
    """BASELINE cfg 2-5 at full size on the device operator: the returned pairs satisfy
    ||A x - lambda x||_2 / |lambda| <= 1e-10 (north star), X^T X = I, X^T A X = diag(eig); the restart-stress
    leg must restart at least twice with locked roots (reference diaglib.f90:1795-1825)."""
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ctx.synth_setup(n, 0, n)
        if guess == "unit":
            ev = _unit_guess_panel(ctx, n, m)
        else:
            ev = ctx.panel(n, m); ctx.fill_guess(ev, 2, support_rows=6000)
        mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
        # reference tolerance semantics (rms < tol, max < 10 tol, diaglib.f90:1741).  At n = 1e7 max|r| of a converged pair
        # stops falling at a rounding floor: 3.4e-13 (2.6e-13 .. 5.0e-13 from iteration to iteration) while no root is locked
        # (tools/floor_probe.py, tests/test_floor_gpu.py), and up to 3.7e-12 for the last two roots once the other thirty are
        # locked and the basis is [X | P | W] with two live columns (measured on the LOBPCG run of this test at tol = 3e-13:
        # max|r| of roots 31 / 32 stays at 3.65e-12 / 3.76e-12 from iteration 19 on while their rms goes on to 1.2e-15 --
        # whether a history ends above or below 3e-12 depends on the last bits of the Rayleigh-Ritz eigenvectors).  The
        # reference's floor is higher still (4.1e-11 at n = 1e6).  tol = 1e-12 keeps 10 tol 2.7 times above the highest floor
        # seen and 30 times above the unlocked one; it is the tolerance of the bench line of this shape.
        # NB the north star's ||A x - lambda x||_2 / |lambda| <= 1e-10 is ASSERTED below, not implied by this tolerance: rms < tol
        # only guarantees rms sqrt(n) / |lambda_min| = 1e-12 * 3162 / 2.9 = 1.1e-9 at n = 1e7.  The rule that ends the solve is
        # max|r| < 10 tol, which trips long after the rms has passed tol: the LOBPCG run of this shape ends at a relative
        # residual of 1.2e-11 (bench line of r04, gpurun_out/r04_cfg5_1.json: max_rel_residual), a margin of 8 under the bound
        # checked here.
        tol = 1e-13 if n < 10_000_000 else 1e-12
        if solver == "davidson":
            eig, _, ok, info = ctx.davidson_driver(n, t, m, 400, tol, max_dav, 0.0, mv, pc, ev)
        else:
            eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 400, tol, 0.0, mv, pc, ev)
        assert ok, info
        if guess != "unit":
            assert info["restarts"] >= 2, info
        assert np.all(np.diff(eig[:t]) > 0)
        ax = ctx.panel(n, m)
        ctx.synth_matvec(ev, ax)
        xt = ev.col(0, t)
        h = ctx.gram(xt, ax.col(0, t))                      # X^T A X = diag(eig)
        assert np.abs(h - np.diag(eig[:t])).max() < 1e-9
        assert np.abs(ctx.gram(xt, xt) - np.eye(t)).max() < 1e-12
        # residual norms through the fused kernel itself: r = AX I - eig * X I
        r = ctx.panel(n, t); e2 = ctx.panel(n, t)
        rn = ctx.ritz_residual(xt, ax.col(0, t), np.eye(t), eig[:t], t, np.zeros(t, np.int32), e2, r)
        rel = rn[0, :] * np.sqrt(n) / np.abs(eig[:t])
        assert rel.max() <= 1e-10, rel
        for p_ in (r, e2, ax, ev):
            p_.free()
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.trim()                       # hand the multi-GB panels back before the next case


def test_restart_stress_cfg5_width_vs_oracle(ctx, oracle):
    """SURVEY 8d restart stress at an oracle-sized n: Davidson, 32 roots, n_max = 37, max_dav = 10, the seed-2 random
    guess, device callbacks -> several restarts with locked roots (reference diaglib.f90:1795-1825 incl. the
    n_rst zero-column quirk, :1696-1702 diagonal patch).  Eigenpairs against the oracle; iteration / restart
    counts within the documented 10 % (random guess: counts depend on last-bit differences of the Gram sums)."""
    n, t, m, max_dav, tol = 6000, 32, 37, 10, 1e-10
    g = oracle.guess_u01(2, n, m)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    try:
        ctx.synth_setup(n, 0, n); oracle.synth_setup(n, 0, n)
        ev = ctx.panel(n, m); ctx.fill_guess(ev, 2)
        assert np.array_equal(ev.download(), g)                    # device generator == oracle generator, bit for bit
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 600, tol, max_dav, 0.0, capi.fn_address("dla_synth_matvec"),
                                               capi.fn_address("dla_synth_precnd"), ev)
        vec = ev.download()
    finally:
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eo, vo, oko, tr = oracle.davidson(n, t, m, 600, tol, max_dav, 0.0, oracle.fn("orc_synth_matvec"),
                                      oracle.fn("orc_synth_precnd"), g)
    assert ok and oko, (info, tr.iters)
    assert info["restarts"] >= 2 and tr.restarts >= 2, (info, tr.restarts)
    assert abs(info["iters"] - tr.iters) <= max(1, tr.iters // 10), (info, tr.iters)
    assert abs(info["restarts"] - tr.restarts) <= 1
    assert np.allclose(eig[:t], eo[:t], rtol=1e-10, atol=0)
    sgn = np.sign((vec[:, :t] * vo[:, :t]).sum(0))
    assert np.abs(vec[:, :t] * sgn - vo[:, :t]).max() < 1e-5
    assert np.abs(vec[:, :t].T @ vec[:, :t] - np.eye(t)).max() < 1e-11


def test_gen_davidson_restart_keeps_metric_block(ctx, oracle, rng):
    """Restart of the generalised Davidson: the kept block's B*x stays in bspace (deliberate fix of the
    reference's zeroing, SURVEY 8a A13 / DESIGN.md) -- the solve converges to the true generalised eigenvalues."""
    import scipy.linalg as sla
    n, t, m = 600, 4, 8
    oracle.dense_setup(n); s = oracle.metric_setup(n)
    mv, pc, bv = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd"), oracle.fn("orc_metric_matvec")
    g = np.asfortranarray(rng.random((n, m)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    eig, vec, ok, info = ctx.gen_david_driver(n, t, m, 400, 1e-8, 5, 0.0, mv, pc, bv, g)
    eo, vo, oko, tr = oracle.gen_davidson(n, t, m, 400, 1e-8, 5, 0.0, mv, pc, bv, g)
    assert ok and oko and info["restarts"] >= 1 and tr.restarts >= 1
    idx = np.arange(1, n + 1.0); a = 1.0 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1.0)
    want = sla.eigh(a, s, eigvals_only=True)[:t]
    assert np.allclose(eig[:t], want, rtol=0, atol=1e-7) and np.allclose(eo[:t], want, rtol=0, atol=1e-7)
    x = vec[:, :t]
    assert np.abs(x.T @ s @ x - np.eye(t)).max() < 1e-8          # B-orthonormal eigenvectors
