#!/usr/bin/env python3
"""The drivers on degenerate problems (host-mode callbacks, small n): operators with multiple / zero / all-equal eigenvalues,
guesses with duplicate or zero columns, blocks wider than the number of distinct directions the operator can produce.
Accepted: converged with residuals below the tolerance and eigenvalues that ARE eigenvalues (checked against numpy), or
ok = False; never ok = True with anything else.

    python tools/fuzz_degenerate_drivers.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
first = int(sys.argv[3]) if len(sys.argv) > 3 else -1
if first < 0:
    # parent: the Fortran drivers end the process on a catastrophic orthogonalisation failure (`error stop`, as the reference
    # does): run the cases in children and go on behind a case that stopped
    import subprocess
    nxt, bad, notok, stopped, ref_ok = 0, 0, 0, 0, 0
    while nxt < cases:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), str(cases), str(seed), str(nxt)], capture_output=True, text=True)
        done = nxt
        for ln in p.stdout.splitlines():
            if ln.startswith("CASE "):
                done = int(ln.split()[1]) + 1
            elif ln.startswith("FAIL"):
                bad += 1; print(ln, flush=True)
            elif ln.startswith("NOTOK"):
                notok += 1
        if p.returncode != 0 and done < cases:
            stopped += 1
            # does the unmodified reference (oracle/_ref, when it has been built) get through this case?
            verdict = "reference not available"
            if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdiaglib_ref.so")) or os.environ.get("FUZZ_REF"):
                env = dict(os.environ, FUZZ_IMPL="reference", FUZZ_ONLY="1")
                pr = subprocess.run([sys.executable, os.path.abspath(__file__), str(cases), str(seed), str(done)], capture_output=True, text=True, env=env)
                line = [ln for ln in pr.stdout.splitlines() if ln.startswith("reference:")]      # (its `stop` ends with exit code 0)
                verdict = "THE REFERENCE GETS THROUGH: " + line[0] if line else "the reference stops too: " + " ".join(pr.stderr.split()[:8])
                if line:
                    ref_ok += 1
            print(f"case {done} stopped the process ({verdict}):", (p.stdout.splitlines() + p.stderr.splitlines())[-3:-2], flush=True)
            done += 1
        nxt = done
    print(f"{cases} cases x 2 solvers, {bad} failures, {notok} not converged / refused, {stopped} stopped with an error "
          f"({ref_ok} of them where the reference gets through)", flush=True)
    sys.exit(1 if bad else 0)
REF = None
if os.environ.get("FUZZ_IMPL") == "reference":
    import ctypes as C
    from oracle.pyoracle import Reference
    REF = Reference()
    MV_T = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double))
    PC_T = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
elif os.environ.get("FUZZ_HOSTSIM"):         # the product's host logic on the host-memory test engine (no GPU needed)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim
    capi.load(hostsim.build())
ctx = capi.Context() if REF is None else None
bad = 0
notok = 0
kinds = ["identity", "zero", "pairs", "band_unit", "dup_guess", "zero_guess_col", "rank_one_plus_shift"]
for it in range(cases):
    rng = np.random.default_rng([seed, it])
    if it < first:
        continue
    kind = kinds[it % len(kinds)]
    n = int(rng.integers(60, 400))
    m = int(rng.integers(2, 12)); t = int(rng.integers(1, m + 1))
    n = max(n, 14 * m)                           # (room for max_dav = 10 blocks and the expansion behind them)
    d = np.arange(1, n + 1, dtype=np.float64)
    a = np.diag(d)
    if kind == "identity":
        a = np.eye(n) * 3.0
    elif kind == "zero":
        a = np.zeros((n, n))
    elif kind == "pairs":
        # every eigenvalue twice, but not a diagonal matrix (with the exact inverse of a diagonal operator as preconditioner the
        # first correction is the iterate itself and what the solvers find is decided by rounding noise -- in the reference too):
        # plane rotations between neighbouring pairs keep the spectrum
        a = np.diag(np.repeat(np.arange(1, n // 2 + 2, dtype=np.float64), 2)[:n])
        cs, sn = np.cos(0.7), np.sin(0.7)
        for i0 in range(1, n - 1, 2):
            g2 = np.eye(n); g2[i0, i0] = cs; g2[i0 + 1, i0 + 1] = cs; g2[i0, i0 + 1] = -sn; g2[i0 + 1, i0] = sn
            a = g2 @ a @ g2.T
        a = 0.5 * (a + a.T)
    elif kind in ("band_unit", "dup_guess", "zero_guess_col"):
        # (a purely diagonal operator with its exact diagonal preconditioner gives Davidson no new direction at all)
        hb = int(rng.integers(1, 5))
        for q in range(1, hb + 1):
            v = 0.3 / q * np.cos(np.arange(n - q) + q)
            a += np.diag(v, q) + np.diag(v, -q)
    elif kind == "rank_one_plus_shift":
        w = rng.standard_normal(n); a = 2.0 * np.eye(n) + np.outer(w, w) / n
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    if kind == "dup_guess" and m > 1:
        g[:, m - 1] = g[:, 0]
    if kind == "zero_guess_col":
        g[:, m - 1] = 0.0
    if kind in ("pairs", "rank_one_plus_shift"):
        g = np.asfortranarray(rng.standard_normal((n, m)))
    dg = np.diag(a).copy()
    mv = lambda x: a @ x
    def pc(fac, x):
        den = dg[:, None] + fac
        return np.where(np.abs(den) > 1e-5, x / np.where(den == 0.0, 1.0, den), x)
    want = np.linalg.eigvalsh(a)
    if REF is not None:
        def mvf(pn, pm, px, pax):
            nn, mm = pn[0], pm[0]
            np.ctypeslib.as_array(pax, (mm, nn)).T[:, :] = mv(np.ctypeslib.as_array(px, (mm, nn)).T)
        def pcf(pn, pm, pf, px, ppx):
            nn, mm = pn[0], pm[0]
            np.ctypeslib.as_array(ppx, (mm, nn)).T[:, :] = pc(pf[0], np.ctypeslib.as_array(px, (mm, nn)).T)
        cmv, cpc = MV_T(mvf), PC_T(pcf)
        pmv, ppc = C.cast(cmv, C.c_void_p).value, C.cast(cpc, C.c_void_p).value
        e1, _, ok1 = REF.davidson(n, t, m, 300, 1e-9, 10, 0.0, pmv, ppc, g.copy(order="F"))
        e2, _, ok2 = REF.lobpcg(n, t, m, 300, 1e-9, 0.0, pmv, ppc, g.copy(order="F"))
        print("reference: davidson ok", ok1, "err", float(np.abs(np.sort(e1[:t]) - want[:t]).max()), "lobpcg ok", ok2, "err", float(np.abs(np.sort(e2[:t]) - want[:t]).max()), flush=True)
        sys.exit(0)
    for solver in ("davidson", "lobpcg"):
        try:
            if solver == "davidson":
                eig, vec, ok, info = ctx.davidson_driver(n, t, m, 300, 1e-9, 10, 0.0, mv, pc, g.copy(order="F"))
            else:
                eig, vec, ok, info = ctx.lobpcg_driver(n, t, m, 300, 1e-9, 0.0, mv, pc, g.copy(order="F"))
        except Exception as e:
            print("NOTOK exception", kind, solver, str(e)[:100], flush=True)
            continue
        if not ok:
            print("NOTOK", kind, solver, flush=True)
            continue
        r = a @ vec[:, :t] - vec[:, :t] * eig[:t]
        res = np.abs(r).max()
        # every reported value is an eigenvalue, and the t lowest are found (multiplicities included)
        miss = np.abs(np.sort(eig[:t]) - want[:t]).max()
        orth = np.abs(vec[:, :t].T @ vec[:, :t] - np.eye(t)).max()
        if not (res < 1e-6 and miss < 1e-6 * max(1.0, abs(want[t - 1])) and orth < 1e-8) or not np.isfinite(res):
            bad += 1
            print("FAIL", dict(kind=kind, solver=solver, n=n, m=m, t=t), dict(res=float(res), miss=float(miss), orth=float(orth), iters=info["iters"]), flush=True)
    print("CASE", it, flush=True)
sys.exit(0)
