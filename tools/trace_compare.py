#!/usr/bin/env python3
"""Print the per-iteration convergence table of the HIP path (verbose=True, the reference's
own table format) next to the oracle's trace for one of the parity cases.  Debug aid."""
import os
import sys

os.environ.setdefault("OMP_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
solver = sys.argv[3] if len(sys.argv) > 3 else "davidson"
T, M = 8, 13
ctx = capi.Context()
o = Oracle()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
o.synth_setup(n, 0, n)
g = np.zeros((n, M), order="F"); g[np.arange(M), np.arange(M)] = 1.0
ev = ctx.panel(g)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
if solver == "davidson":
    eig, _, ok, info = ctx.davidson_driver(n, T, M, 100, tol, 20, 0.0, mv, pc, ev, verbose=True)
    eo, vo, oko, tr = o.davidson(n, T, M, 100, tol, 20, 0.0, o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd"), g)
else:
    eig, _, ok, info = ctx.lobpcg_driver(n, T, M, 100, tol, 0.0, mv, pc, ev, verbose=True)
    eo, vo, oko, tr = o.lobpcg(n, T, M, 100, tol, 0.0, o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd"), g)
sys.stdout.flush()
np.set_printoptions(linewidth=220, precision=4)
print("HIP   :", info, ok)
print("oracle:", tr.iters, tr.matvec_cols, list(tr.n_act))
for it in range(tr.iters):
    print("oracle it", it + 1, "rms ", tr.rms[it])
    print("oracle it", it + 1, "rmax", tr.rmax[it])
