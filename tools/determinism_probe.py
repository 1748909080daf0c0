#!/usr/bin/env python3
"""Run the benchmark solve repeatedly and compare eigenvalues and eigenvectors bit for bit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from diaglib_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
t, m = 8, 13
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
ev = ctx.panel(g)
ref = None
same = True
for solver in ("davidson", "lobpcg"):
    ref = None
    for r in range(reps):
        ev.upload(g)
        if solver == "davidson":
            eig, _, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        else:
            eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 200, 2e-13, 0.0, mv, pc, ev)
        v = ev.download()
        if ref is None:
            ref = (eig.copy(), v, info["iters"])
        else:
            if not (np.array_equal(eig, ref[0]) and np.array_equal(v, ref[1]) and info["iters"] == ref[2]):
                same = False
                print(solver, "run", r, "differs: max eig diff", np.abs(eig - ref[0]).max(), "max vec diff", np.abs(v - ref[1]).max(), info["iters"], ref[2])
    print(solver, "iterations", ref[2], "bitwise identical over", reps, "runs:", same, flush=True)
sys.exit(0 if same else 1)
