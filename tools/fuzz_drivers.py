#!/usr/bin/env python3
"""Randomised driver runs (Davidson, LOBPCG, generalised variants) on random dense symmetric problems with Python
host callbacks, against numpy/scipy dense solutions.   python tools/fuzz_drivers.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import scipy.linalg as sl
from diaglib_amd import capi

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
bad = 0
for it in range(cases):
    n = int(rng.integers(120, 1500)); n += int(rng.integers(0, 2))
    t = int(rng.integers(1, 12)); m = min(2 * t, t + 5)
    solver = rng.choice(["davidson", "lobpcg", "gen_david", "lobpcg_gen"])
    guess = rng.choice(["unit", "rand", "zero"])
    if "lobpcg" in solver:
        # LOBPCG with the reference's preconditioner (a_ii - eig(1))^-1 stagnates from a random start on these spectra,
        # in the reference itself as well (seed 2: n=1309, one root, reference returns ok=F after 400 iterations)
        guess = "unit"
    d = np.sort(rng.random(n)) * n * 0.2 + np.arange(n) * 0.5 + 1.0
    q = rng.standard_normal((n, 6)) * 0.3
    a = np.diag(d) + q @ q.T
    dg = np.diag(a).copy()
    b = None
    if "gen" in solver:
        gq = rng.standard_normal((n, 4)) * 0.2
        b = np.eye(n) + gq @ gq.T
        want = sl.eigh(a, b, eigvals_only=True)[:t]
    else:
        want = np.linalg.eigvalsh(a)[:t]
    if guess == "unit":
        g = np.zeros((n, m), order="F"); g[np.argsort(dg)[:m], np.arange(m)] = 1.0
    elif guess == "rand":
        g = np.asfortranarray(rng.random((n, m)) - 0.5)
    else:
        g = np.zeros((n, m), order="F")
    mv = lambda x: a @ x
    pc = lambda fac, x: np.where(np.abs(dg + fac)[:, None] > 1e-5, x / (dg + fac)[:, None], x)
    bv = (lambda x: b @ x) if b is not None else None
    try:
        if solver == "davidson":
            e, v, ok, info = ctx.davidson_driver(n, t, m, 400, 1e-9, 20, 0.0, mv, pc, g)
        elif solver == "lobpcg":
            e, v, ok, info = ctx.lobpcg_driver(n, t, m, 400, 1e-9, 0.0, mv, pc, g)
        elif solver == "gen_david":
            e, v, ok, info = ctx.gen_david_driver(n, t, m, 400, 1e-9, 20, 0.0, mv, pc, bv, g)
        else:
            e, v, ok, info = ctx.lobpcg_driver(n, t, m, 400, 1e-9, 0.0, mv, pc, g, bvec=bv)
        err = np.abs(e[:t] - want).max() / max(1.0, np.abs(want).max())
        good = ok and err < 1e-8
    except Exception as ex:   # noqa: BLE001
        good, err, info = False, float("nan"), str(ex)
    if not good:
        bad += 1
        print("FAIL", dict(n=n, t=t, m=m, solver=str(solver), guess=str(guess)), "err", err, info, flush=True)
print(f"{cases} driver cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
