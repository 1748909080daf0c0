#!/usr/bin/env python3
"""Iteration counts of seeded random-guess solves on the three implementations -- HIP path, oracle (C restatement), unmodified
reference (oracle/_ref) -- side by side.  Round-5 review: with random guesses the histories agree only statistically (the counts
depend on last-bit differences of the Gram sums; the tests allow +-10-15 %), so a real regression of that size could hide in the
tolerance.  On one GPU the HIP counts are deterministic (fixed reduction shapes): this record pins them per round
(profiles/rNN/iteration_counts.txt); a later build whose counts move shows up in the diff.
    python tools/iteration_record.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle, Reference  # noqa: E402

ctx, o = capi.Context(), Oracle()
ref = Reference() if Reference.available() else None
print("# solver, n, roots, n_max, max_dav, tol, guess seed | iterations (matvec columns): hip / oracle / reference", flush=True)
for solver, n, t, m, max_dav, tol, seed in (("davidson", 2000, 4, 8, 20, 1e-8, 5), ("lobpcg", 2000, 4, 8, 20, 1e-8, 5),
                                            ("davidson", 1000, 10, 15, 20, 1e-8, 7), ("davidson", 600, 4, 8, 10, 1e-8, 9),
                                            ("lobpcg", 3000, 16, 21, 20, 1e-8, 11), ("davidson", 3000, 16, 21, 10, 1e-9, 13),
                                            ("lobpcg", 2500, 32, 37, 20, 1e-8, 15), ("davidson", 2500, 32, 37, 10, 1e-8, 17)):
    o.dense_setup(n)
    mv, pc = o.fn("orc_dense_matvec"), o.fn("orc_dense_precnd")
    g = np.asfortranarray(np.random.default_rng(seed).random((n, m)) - 0.5)
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    if solver == "davidson":
        e, _, ok, info = ctx.davidson_driver(n, t, m, 400, tol, max_dav, 0.0, mv, pc, g.copy(order="F"))
        eo, _, oko, tr = o.davidson(n, t, m, 400, tol, max_dav, 0.0, mv, pc, g)
    else:
        e, _, ok, info = ctx.lobpcg_driver(n, t, m, 400, tol, 0.0, mv, pc, g.copy(order="F"))
        eo, _, oko, tr = o.lobpcg(n, t, m, 400, tol, 0.0, mv, pc, g)
    rtxt = "-"
    if ref is not None and solver == "davidson":
        o.synth_counters(reset=True)
        er, _, okr = ref.davidson(n, t, m, 400, tol, max_dav, 0.0, mv, pc, g)
        rtxt = f"ok={okr} eig diff {np.abs(er[:t] - e[:t]).max():.1e}"
    print(f"{solver:9s} n={n} roots={t} n_max={m} max_dav={max_dav} tol={tol:g} seed={seed} | hip {info['iters']} ({info['matvec_cols']}) ok={ok} | "
          f"oracle {tr.iters} ({tr.matvec_cols}) ok={oko} eig diff {np.abs(eo[:t] - e[:t]).max():.1e} | reference {rtxt}", flush=True)
# the benchmark's own random-guess leg (device callbacks, seed-2 guess on the leading 2000 rows)
n, t, m = 2_000_000, 8, 13
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
ev = ctx.panel(n, m)
ctx.fill_guess(ev, 2, 2000)
e, _, ok, info = ctx.davidson_driver(n, t, m, 400, 2e-13, 20, 0.0, capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd"), ev)
print(f"davidson  n={n} roots={t} n_max={m} benchmark operator, seed-2 guess on 2000 rows | hip {info['iters']} ({info['matvec_cols']}), {info['restarts']} restarts ok={ok}", flush=True)
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
