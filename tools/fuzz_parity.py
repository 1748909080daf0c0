#!/usr/bin/env python3
"""Randomised PARITY runs: the HIP drivers against the oracle restatement (iteration counts, eigenvalues, ok) and -- when
oracle/_ref has been built -- against the unmodified reference (eigenvalues, ok), on random dense symmetric problems with
host callbacks: block widths 1 .. 40, few Davidson blocks (restarts, the n_rst rule), shifts, loose and tight tolerances,
unit and random guesses, Davidson / LOBPCG / their generalised variants.

    python tools/fuzz_parity.py [cases] [seed] [only this case]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle, Reference  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if os.environ.get("FUZZ_HOSTSIM"):           # the product's host logic on the host-memory test engine (no GPU needed)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim
    capi.load(hostsim.build())
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
orc = Oracle()
ref = Reference() if Reference.available() else None
MV_T = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double))
PC_T = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
bad = 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run this case alone
for it in range(cases):
    if only >= 0 and it != only:
        continue
    rng = np.random.default_rng([seed, it])
    t = int(rng.choice([rng.integers(1, 9), rng.integers(9, 25)]))
    m = int(min(t + rng.integers(0, 16), 2 * t + 3, 40))
    if os.environ.get("FUZZ_WIDE"):              # blocks beyond the 48 columns the fused sweeps and the device chains take
        t = int(rng.integers(30, 70)); m = int(min(t + rng.integers(0, 20), 90))
    n = int(rng.integers(max(300, 30 * m), max(1800, 32 * m))); n += int(rng.integers(0, 2))
    if rng.random() < 0.3:
        ctx.set_option(capi.OPT_STAGE_CHUNKS, int(rng.integers(2, 5)))      # host-mode callbacks in column chunks
    else:
        ctx.set_option(capi.OPT_STAGE_CHUNKS, 0)
    solver = str(rng.choice(["davidson", "davidson", "lobpcg", "gen_david", "lobpcg_gen"]))
    max_dav = int(rng.choice([2, 3, 5, 10, 20]))
    tol = float(rng.choice([1e-6, 1e-9, 1e-11]))
    shift = float(rng.choice([0.0, 0.0, 1.5]))
    guess = "unit" if "lobpcg" in solver else str(rng.choice(["unit", "rand"]))
    d = np.sort(rng.random(n)) * n * 0.2 + np.arange(n) * 0.5 + 1.0
    q = rng.standard_normal((n, 6)) * 0.3
    a = np.diag(d) + q @ q.T
    dg = np.diag(a).copy()
    b = None
    if "gen" in solver:
        gq = rng.standard_normal((n, 4)) * 0.2
        b = np.eye(n) + gq @ gq.T
    if guess == "unit":
        g = np.zeros((n, m), order="F"); g[np.argsort(dg)[:m], np.arange(m)] = 1.0
    else:
        g = np.asfortranarray(rng.random((n, m)) - 0.5)
    mv = lambda x: a @ x + shift * (x if b is None else b @ x)        # the harness convention: the operator carries the shift
    pc = lambda fac, x: np.where(np.abs(dg + shift + fac)[:, None] > 1e-5, x / (dg + shift + fac)[:, None], x)
    bv = (lambda x: b @ x) if b is not None else None

    def mvf(pn, pm, px, pax):
        nn, mm = pn[0], pm[0]
        np.ctypeslib.as_array(pax, (mm, nn)).T[:, :] = mv(np.ctypeslib.as_array(px, (mm, nn)).T)

    def bvf(pn, pm, px, pax):
        nn, mm = pn[0], pm[0]
        np.ctypeslib.as_array(pax, (mm, nn)).T[:, :] = bv(np.ctypeslib.as_array(px, (mm, nn)).T)

    def pcf(pn, pm, pf, px, ppx):
        nn, mm = pn[0], pm[0]
        np.ctypeslib.as_array(ppx, (mm, nn)).T[:, :] = pc(pf[0], np.ctypeslib.as_array(px, (mm, nn)).T)
    cmv, cpc, cbv = MV_T(mvf), PC_T(pcf), MV_T(bvf)
    pmv, ppc, pbv = (C.cast(f, C.c_void_p).value for f in (cmv, cpc, cbv))
    spec = dict(n=n, t=t, m=m, solver=solver, max_dav=max_dav, tol=tol, shift=shift, guess=guess)
    # every case also with the pending blocks switched off (DLA_OPT_PENDING_BLOCKS = 0: each block finished in memory) and with the
    # three-pass schedule forced from the first chain (tune knob 6 = 13), in this process: same ok, same eigenvalues, same history
    variants = {}
    try:
        for vname, (pend, knob) in (("no_pending", (0, 0)), ("three_pass", (1, 13)), ("five_sweep", (1, 12))):
            ctx.set_option(capi.OPT_PENDING_BLOCKS, pend); ctx.set_option(100 + 6, knob)
            if solver == "davidson":
                variants[vname] = ctx.davidson_driver(n, t, m, 300, tol, max_dav, shift, mv, pc, g.copy(order="F"))
            elif solver == "lobpcg":
                variants[vname] = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, g.copy(order="F"))
            elif solver == "gen_david":
                variants[vname] = ctx.gen_david_driver(n, t, m, 300, tol, max_dav, shift, mv, pc, bv, g.copy(order="F"))
            else:
                variants[vname] = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, g.copy(order="F"), bvec=bv)
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print("FAIL (exception in a variant)", spec, str(ex)[:200], flush=True)
        continue
    finally:
        ctx.set_option(capi.OPT_PENDING_BLOCKS, 1); ctx.set_option(100 + 6, 0)
    for kv in filter(None, os.environ.get("FUZZ_TUNE", "").split(",")):       # e.g. FUZZ_TUNE=6=3: the main run on the host-driven loops
        ctx.set_option(100 + int(kv.split("=")[0]), int(kv.split("=")[1]))
    try:
        if solver == "davidson":
            e, v, ok, info = ctx.davidson_driver(n, t, m, 300, tol, max_dav, shift, mv, pc, g)
            eo, vo, oko, tr = orc.davidson(n, t, m, 300, tol, max_dav, shift, pmv, ppc, g)
            er, okr = (ref.davidson(n, t, m, 300, tol, max_dav, shift, pmv, ppc, g)[::2] if ref else (eo, oko))
        elif solver == "lobpcg":
            e, v, ok, info = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, g)
            eo, vo, oko, tr = orc.lobpcg(n, t, m, 300, tol, shift, pmv, ppc, g)
            er, okr = (ref.lobpcg(n, t, m, 300, tol, shift, pmv, ppc, g)[::2] if ref else (eo, oko))
        elif solver == "gen_david":
            e, v, ok, info = ctx.gen_david_driver(n, t, m, 300, tol, max_dav, shift, mv, pc, bv, g)
            eo, vo, oko, tr = orc.gen_davidson(n, t, m, 300, tol, max_dav, shift, pmv, ppc, pbv, g)
            er, okr = eo, oko          # (the reference's gen_david_driver has the defect recorded in DESIGN: the oracle is the arbiter)
        else:
            e, v, ok, info = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, g, bvec=bv)
            eo, vo, oko, tr = orc.lobpcg_gen(n, t, m, 300, tol, shift, pmv, ppc, pbv, g)
            er, okr = (ref.lobpcg(n, t, m, 300, tol, shift, pmv, ppc, g, bvec=pbv)[::2] if ref else (eo, oko))
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print("FAIL (exception)", spec, str(ex)[:200], flush=True)
        continue
    scale = max(1.0, np.abs(eo[:t]).max())
    res = dict(ok=(ok, oko, okr), d_oracle=float(np.abs(e[:t] - eo[:t]).max() / scale), d_ref=float(np.abs(e[:t] - er[:t]).max() / scale),
               iters=(info["iters"], tr.iters))
    # unit guesses: the history is robust to rounding; random guesses and tolerances near the rounding floor of max|r| (1e-11 on
    # these spectra: docs/HISTORY.md 11.7) end a few sweeps earlier or later with the last bits of the small eigensolver
    slack = 1 if (guess == "unit" and tol >= 1e-10) else max(3, tr.iters // 8)
    if "lobpcg" in solver and slack == 1:
        # LOBPCG's 3m x 3m Rayleigh-Ritz problem has clusters whose eigenvectors the partial solver here and the oracle's solver
        # rotate differently; 2 of 120 unit-guess cases (seed 300: cases 27 and 84) take 13 sweeps against 11 -- on the round-4
        # build as well, with every schedule, eigenvalues equal to 1e-13
        slack = 2
    if "lobpcg" in solver and (m > 24 or m <= t + 1):   # (wide LOBPCG blocks, or at most one guard vector behind the wanted roots: the order in
        slack = max(3, tr.iters // 6)               #  which the last roots lock moves the count by 10 %, with or without the pending factor)
    lim = max(1e-9, 50.0 * tol * tol)          # eigenvalue error ~ residual^2; both sides stop anywhere below tol
    good = ok == oko == okr and (not ok or (res["d_oracle"] < lim and res["d_ref"] < lim)) and abs(info["iters"] - tr.iters) <= slack
    for vname, (ev_, _, okv, infov) in variants.items():
        dv = float(np.abs(ev_[:t] - e[:t]).max() / scale)
        res[vname] = (bool(okv), dv, infov["iters"])
        good = good and okv == ok and (not ok or dv < max(1e-10, lim)) and abs(infov["iters"] - info["iters"]) <= max(1, slack)
    if not good:
        bad += 1
        print("FAIL", dict(spec, case=it), res, flush=True)
    elif only >= 0:
        print("ok  ", dict(spec, case=it), res, flush=True)
print(f"{cases} parity cases, {bad} failures (reference {'used' if ref else 'not available'})", flush=True)
sys.exit(1 if bad else 0)
