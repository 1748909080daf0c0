#!/usr/bin/env python3
"""Randomised parity runs of the linear-response drivers (caslr_eff_driver, caslr_driver) against the unmodified reference
(oracle/_ref) on the oracle's sample LR operators: sizes, block widths, subspace caps, tolerances, unit and random guesses.
Compared: ok, the eigenvalues (against the reference and against the dense solution of the 2n x 2n pencil).

    python tools/fuzz_parity_lr.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.linalg as sl  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle, Reference  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if os.environ.get("FUZZ_HOSTSIM"):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hostsim
    capi.load(hostsim.build())
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
o = Oracle()
ref = Reference() if Reference.available() else None
bad = 0
for it in range(cases):
    rng = np.random.default_rng([seed, it])
    n = int(rng.integers(120, 500))
    t = int(rng.integers(1, 7)); m = int(t + rng.integers(0, 6))
    max_dav = int(rng.choice([5, 10, 20])); tol = float(rng.choice([1e-6, 1e-8, 1e-10]))
    n = max(n, m * (max_dav + 1) + 10)           # (the expansion spaces live in R^n: room for max_dav blocks and one more)
    trad = bool(rng.integers(0, 2)); guess = str(rng.choice(["unit", "rand"]))
    apb, amb, spd, smd = o.lr_setup(n)
    fn = [o.fn(k) for k in ("orc_lr_apb", "orc_lr_amb", "orc_lr_spd", "orc_lr_smd", "orc_lr_prec1" if trad else "orc_lr_prec")]
    if guess == "unit":
        g = np.zeros((2 * n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    else:
        g = np.asfortranarray(rng.random((2 * n, m)) - 0.5)
    spec = dict(case=it, n=n, t=t, m=m, max_dav=max_dav, tol=tol, driver="caslr" if trad else "caslr_eff", guess=guess)
    if os.environ.get("FUZZ_VERBOSE"):
        print("case", spec, flush=True)
    try:
        ctx.set_option(capi.OPT_CASLR_ALGORITHM, 0)
        solve = ctx.caslr_driver if trad else ctx.caslr_eff_driver
        e, v, ok, info = solve(n, t, m, 300, tol, max_dav, *fn, g)
        if ref:
            er, _, okr = (ref.caslr if trad else ref.caslr_eff)(n, t, m, 300, tol, max_dav, *fn, g)
        else:
            er, okr = e, ok
    except Exception as ex:   # noqa: BLE001
        bad += 1; print("FAIL (exception)", spec, str(ex)[:200], flush=True); continue
    a, b, s, d = 0.5 * (apb + amb), 0.5 * (apb - amb), 0.5 * (spd + smd), 0.5 * (spd - smd)
    w = sl.eigvals(np.block([[a, b], [b, a]]), np.block([[s, d], [-d, -s]])).real
    want = np.sort(w[w > 0])[:t]
    res = dict(ok=(ok, okr), d_ref=float(np.abs(e[:t] - er[:t]).max()), d_dense=float(np.abs(e[:t] - want).max()), iters=info["iters"])
    lim = max(1e-8, 100.0 * tol * tol) * max(1.0, float(np.abs(want).max()))
    if not (ok == okr and (not ok or (res["d_ref"] < lim and res["d_dense"] < lim))):
        bad += 1; print("FAIL", spec, res, flush=True)
print(f"{cases} linear-response parity cases, {bad} failures (reference {'used' if ref else 'not available'})", flush=True)
sys.exit(1 if bad else 0)
