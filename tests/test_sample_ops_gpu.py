"""GPU: the device-resident sample operators of the linear-response and generalised drivers (dla_synth_apbmul / ambmul /
spdmul / smdmul / metric / lrprec1 / lrprec2: matrix-free y = d x + W C W^T x around the benchmark operator's W) and the
drivers running on them with DEVICE callbacks -- the mode the headline benchmark uses for davidson_driver.

 1. every operator against its dense definition (W from the oracle's generator, bit-identical to the device's);
 2. n = 400: caslr_eff_driver / caslr_driver / gen_david_driver / lobpcg_driver(gen_eig) with device callbacks against the
    dense solution of the same problem (scipy; reference diaglib.f90:1024-1481, 558-1022, 1855-2250, 171-556);
 3. n = 2e6 (the benchmark's size): the same solves, checked through residuals and (bi-)orthonormality formed with the
    library's own sweeps -- properties that do not need a dense matrix."""
import numpy as np
import pytest

from diaglib_amd import capi

pytestmark = pytest.mark.gpu
SIGMA, TAU = 0.5, 0.05
J = np.array([[0.0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 0, 1], [0, 0, -1, 0]])


def _dense(oracle, n):
    oracle.synth_setup(n, 0, n)
    w = oracle.synth_w()
    i = np.arange(1, n + 1.0)
    s = 1.0 + 0.5 / (1.0 + (np.arange(1, n + 1) % 7))
    ww = w @ w.T
    ops = dict(a=np.diag(i + 1) + SIGMA * ww, apb=np.diag(i + 5) + SIGMA * ww, amb=np.diag(i + 2) + 0.2 * SIGMA * ww,
               spd=np.diag(s) + TAU * (w @ J @ w.T), smd=np.diag(s) - TAU * (w @ J @ w.T), metric=np.diag(s) + 0.1 * ww)
    return w, s, ops


def _apply(ctx, name, x):
    px, py = ctx.panel(x), ctx.panel(x.shape[0], x.shape[1])
    ctx._chk(ctx.lib.dla_call_matvec(ctx.h, capi.fn_address(name), x.shape[0], x.shape[1], px.ptr, py.ptr))
    return py.download()


@pytest.fixture()
def dev(ctx):
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    yield ctx
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)


def test_sample_operators_match_their_dense_definitions(dev, oracle, rng):
    n, m = 1000, 7
    w, s, ops = _dense(oracle, n)
    dev.synth_setup(n, 0, n)
    x = np.asfortranarray(rng.standard_normal((n, m)))
    for name, key in [("dla_synth_matvec", "a"), ("dla_synth_apbmul", "apb"), ("dla_synth_ambmul", "amb"), ("dla_synth_spdmul", "spd"),
                      ("dla_synth_smdmul", "smd"), ("dla_synth_metric", "metric")]:
        got = _apply(dev, name, x)
        want = ops[key] @ x
        assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max(), name
    assert np.abs(ops["spd"] - ops["smd"].T).max() < 1e-15            # S symmetric, D antisymmetric
    # lrprec_1 / lrprec_2 of the harness (main.f90:234-281) on the operators' diagonals
    aa = 0.5 * (np.diag(ops["apb"]) + np.diag(ops["amb"]))
    xp, xm = x, np.asfortranarray(rng.standard_normal((n, m)))
    for variant, fac in [(1, 0.37), (2, 2.5)]:
        pxp, pxm, pyp, pym = dev.panel(xp), dev.panel(xm), dev.panel(n, m), dev.panel(n, m)
        dev._chk(dev.lib.dla_call_lrprec(dev.h, capi.fn_address(f"dla_synth_lrprec{variant}"), n, m, fac, pxp.ptr, pxm.ptr, pyp.ptr, pym.ptr))
        if variant == 1:
            den = -1.0 / (aa ** 2 - fac ** 2 * s ** 2)
            wp, wm = den[:, None] * (aa[:, None] * xp + fac * s[:, None] * xm), den[:, None] * (aa[:, None] * xm + fac * s[:, None] * xp)
        else:
            den = 1.0 / (fac ** 2 * aa ** 2 - s ** 2)
            wp, wm = den[:, None] * (fac * aa[:, None] * xp + s[:, None] * xm), den[:, None] * (fac * aa[:, None] * xm + s[:, None] * xp)
        assert np.abs(pyp.download() - wp).max() <= 1e-13 * np.abs(wp).max()
        assert np.abs(pym.download() - wm).max() <= 1e-13 * np.abs(wm).max()


def _lr_fns(trad):
    return [capi.fn_address(k) for k in ("dla_synth_apbmul", "dla_synth_ambmul", "dla_synth_spdmul", "dla_synth_smdmul",
                                         "dla_synth_lrprec1" if trad else "dla_synth_lrprec2")]


@pytest.mark.parametrize("trad", [False, True])
def test_lr_drivers_on_device_operators_vs_dense(dev, oracle, trad):
    import scipy.linalg as sla
    n, t, m = 400, 4, 8
    _, _, ops = _dense(oracle, n)
    a, b = 0.5 * (ops["apb"] + ops["amb"]), 0.5 * (ops["apb"] - ops["amb"])
    s, d = 0.5 * (ops["spd"] + ops["smd"]), 0.5 * (ops["spd"] - ops["smd"])
    big = np.block([[a, b], [b, a]]); met = np.block([[s, d], [-d, -s]])
    wd = np.sort(np.real(sla.eig(big, met, right=False)))
    want = wd[wd > 0][:t]
    dev.synth_setup(n, 0, n)
    g = np.zeros((2 * n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    ev = dev.panel(g)
    solve = dev.caslr_driver if trad else dev.caslr_eff_driver
    eig, _, ok, info = solve(n, t, m, 200, 1e-9, 10, *_lr_fns(trad), ev)
    assert ok, info
    assert np.allclose(eig[:t], want, rtol=1e-9, atol=0), (eig[:t], want)
    v = ev.download()[:, :t]
    for i in range(t):
        r = big @ v[:, i] - eig[i] * (met @ v[:, i])
        assert np.linalg.norm(r) / np.linalg.norm(big @ v[:, i]) < 1e-7


@pytest.mark.parametrize("solver", ["gen_davidson", "lobpcg"])
def test_generalised_drivers_on_device_metric_vs_dense(dev, oracle, solver):
    import scipy.linalg as sla
    n, t, m = 400, 4, 8
    _, _, ops = _dense(oracle, n)
    want = sla.eigh(ops["a"], ops["metric"], eigvals_only=True)[:t]
    dev.synth_setup(n, 0, n)
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    ev = dev.panel(g)
    mv, pc, bv = (capi.fn_address(k) for k in ("dla_synth_matvec", "dla_synth_precnd", "dla_synth_metric"))
    if solver == "gen_davidson":
        eig, _, ok, info = dev.gen_david_driver(n, t, m, 200, 1e-10, 10, 0.0, mv, pc, bv, ev)
    else:
        eig, _, ok, info = dev.lobpcg_driver(n, t, m, 200, 1e-10, 0.0, mv, pc, ev, bvec=bv)
    assert ok, info
    assert np.allclose(eig[:t], want, rtol=1e-10, atol=0), (eig[:t], want)
    x = ev.download()[:, :t]
    assert np.abs(x.T @ ops["metric"] @ x - np.eye(t)).max() < 1e-9          # B-orthonormal


def _cols(p, j0, k):
    return p.col(j0, k)


@pytest.mark.parametrize("trad", [False, True])
def test_full_size_linear_response_residuals(dev, trad):
    """n = 2e6 (4e6-dimensional pencil), 4 roots: with v+ = Y + Z, v- = Y - Z the eigen-equations read
    (A+B) v+ = w (S-D) v-  and  (A-B) v- = w (S+D) v+  (reference diaglib.f90:1027-1046); residuals through the library's
    own operators and sweeps."""
    n, t, m = 2_000_000, 4, 8
    dev.synth_setup(n, 0, n)
    try:
        ev = dev.panel(2 * n, m).zero()
        top = np.eye(m, order="F")
        for j in range(m):
            dev._chk(dev.lib.dla_upload(dev.h, ev.ptr + 8 * 2 * n * j, top[:, j].ctypes.data, 8 * m))
        solve = dev.caslr_driver if trad else dev.caslr_eff_driver
        eig, _, ok, info = solve(n, t, m, 200, 1e-9, 10, *_lr_fns(trad), ev)
        assert ok, info
        assert np.all(np.diff(eig[:t]) > 0) and eig[0] > 0
        # Y and Z are the upper / lower halves of every column of the 2n x m block: gather them as n x t panels
        y, z = dev.panel(n, t), dev.panel(n, t)
        for j in range(t):
            dev._chk(dev.lib.dla_copy(dev.h, y.ptr + 8 * n * j, ev.ptr + 8 * 2 * n * j, 8 * n))
            dev._chk(dev.lib.dla_copy(dev.h, z.ptr + 8 * n * j, ev.ptr + 8 * 2 * n * j + 8 * n, 8 * n))
        vp, vm = dev.panel(n, t), dev.panel(n, t)
        dev._chk(dev.lib.dla_copy(dev.h, vp.ptr, y.ptr, 8 * n * t)); dev._chk(dev.lib.dla_axpy(dev.h, n * t, 1.0, z.ptr, vp.ptr))
        dev._chk(dev.lib.dla_copy(dev.h, vm.ptr, y.ptr, 8 * n * t)); dev._chk(dev.lib.dla_axpy(dev.h, n * t, -1.0, z.ptr, vm.ptr))
        l1, r1, l2, r2 = (dev.panel(n, t) for _ in range(4))
        for fn, src, dst in [("dla_synth_apbmul", vp, l1), ("dla_synth_smdmul", vm, r1), ("dla_synth_ambmul", vm, l2), ("dla_synth_spdmul", vp, r2)]:
            dev._chk(dev.lib.dla_call_matvec(dev.h, capi.fn_address(fn), n, t, src.ptr, dst.ptr))
        for lhs, rhs in [(l1, r1), (l2, r2)]:
            num = np.zeros(t); den = np.zeros(t)
            for j in range(t):
                lj, rj = lhs.col(j, 1), rhs.col(j, 1)
                dev._chk(dev.lib.dla_axpy(dev.h, n, -float(eig[j]), rj.ptr, lj.ptr))      # lhs - w rhs
                num[j] = np.sqrt(dev.gram(lj, lj)[0, 0]); den[j] = float(eig[j]) * np.sqrt(dev.gram(rj, rj)[0, 0])
            assert (num / den).max() < 1e-6, num / den
        for p_ in (ev, y, z, vp, vm, l1, r1, l2, r2):
            p_.free()
    finally:
        dev.trim()


@pytest.mark.parametrize("solver", ["gen_davidson", "lobpcg"])
def test_full_size_generalised_residuals(dev, solver):
    """n = 2e6, 8 roots, SPD device metric: ||A x - lambda B x|| / (|lambda| ||B x||) and X^T B X = I through the library's sweeps."""
    n, t, m = 2_000_000, 8, 13
    dev.synth_setup(n, 0, n)
    try:
        ev = dev.panel(n, m).zero()
        top = np.eye(m, order="F")
        for j in range(m):
            dev._chk(dev.lib.dla_upload(dev.h, ev.ptr + 8 * n * j, top[:, j].ctypes.data, 8 * m))
        mv, pc, bv = (capi.fn_address(k) for k in ("dla_synth_matvec", "dla_synth_precnd", "dla_synth_metric"))
        if solver == "gen_davidson":
            eig, _, ok, info = dev.gen_david_driver(n, t, m, 200, 1e-12, 20, 0.0, mv, pc, bv, ev)
        else:
            eig, _, ok, info = dev.lobpcg_driver(n, t, m, 200, 1e-12, 0.0, mv, pc, ev, bvec=bv)
        assert ok, info
        assert np.all(np.diff(eig[:t]) > 0)
        xt = ev.col(0, t)
        ax, bx = dev.panel(n, t), dev.panel(n, t)
        dev._chk(dev.lib.dla_call_matvec(dev.h, mv, n, t, xt.ptr, ax.ptr))
        dev._chk(dev.lib.dla_call_matvec(dev.h, bv, n, t, xt.ptr, bx.ptr))
        assert np.abs(dev.gram(xt, bx) - np.eye(t)).max() < 1e-10                 # B-orthonormal
        assert np.abs(dev.gram(xt, ax) - np.diag(eig[:t])).max() < 1e-8           # X^T A X = diag(eig)
        rel = np.zeros(t)
        for j in range(t):
            aj, bj = ax.col(j, 1), bx.col(j, 1)
            dev._chk(dev.lib.dla_axpy(dev.h, n, -float(eig[j]), bj.ptr, aj.ptr))
            rel[j] = np.sqrt(dev.gram(aj, aj)[0, 0]) / (abs(float(eig[j])) * np.sqrt(dev.gram(bj, bj)[0, 0]))
        assert rel.max() < 1e-8, rel
        for p_ in (ev, ax, bx):
            p_.free()
    finally:
        dev.trim()


def test_a_sample_operator_used_before_its_setup_returns_an_error():
    """Round-5 review: the sample-operator callbacks (reference callback shape: void, no status) used to abort() the process -- which
    holds the GPU -- when they failed.  They now leave their failure to the trampoline that called them (dla_call_matvec /
    dla_call_precnd), which returns it as its own status.  A fresh host thread has no operator set up (the setup belongs to the
    calling thread); the process survives, the context stays usable."""
    import threading
    seen = {}

    def work():
        c = capi.Context()
        try:
            c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
            x, y = c.panel(np.ones((64, 2), order="F")), c.panel(64, 2)
            for name, call in (("matvec", lambda: c.synth_matvec(x, y)),
                               ("spmm", lambda: c._chk(c.lib.dla_call_matvec(c.h, capi.fn_address("dla_spmm_matvec"), 64, 2, x.ptr, y.ptr)))):
                try:
                    call()
                    seen[name] = "no error"
                except capi.DlaError as e:
                    seen[name] = str(e)
            c.synth_setup(64, 0, 64)                 # ... and the same context works once the operator exists
            c.synth_matvec(x, y)
            seen["after"] = float(np.abs(y.download()).max())
        finally:
            c.destroy()

    t = threading.Thread(target=work)
    t.start(); t.join()
    assert "before dla_synth_setup" in seen["matvec"] and "before dla_spmm_setup_csr" in seen["spmm"], seen
    assert seen["after"] > 0.0


def test_the_oracles_restatement_of_the_sample_metric_equals_the_device_operator(dev, oracle, rng):
    """The multi-rank parity tests run the generalised drivers with dla_synth_metric on the device and orc_synth_metric in the oracle
    (oracle/oracle_ops.c, r06): the two must be the same operator, entry by entry (same generator, same order of additions per row up
    to the 4-term sum)."""
    import ctypes as C
    n, m = 5000, 7
    dev.synth_setup(n, 0, n); oracle.synth_setup(n, 0, n)
    x = np.asfortranarray(rng.standard_normal((n, m)))
    y_dev = _apply(dev, "dla_synth_metric", x)
    y_orc = np.zeros((n, m), order="F")
    nn, mm = C.c_int(n), C.c_int(m)
    oracle.lib.orc_synth_metric(C.byref(nn), C.byref(mm), x.ctypes.data_as(C.POINTER(C.c_double)), y_orc.ctypes.data_as(C.POINTER(C.c_double)))
    assert np.abs(y_dev - y_orc).max() <= 8 * np.finfo(float).eps * np.abs(y_orc).max()
    # ... and the metric is symmetric positive definite: x^T B x > 0, x^T B y = y^T B x
    g = x.T @ y_dev
    assert np.abs(g - g.T).max() < 1e-10 * np.abs(g).max() and np.all(np.linalg.eigvalsh(0.5 * (g + g.T)) > 0)
