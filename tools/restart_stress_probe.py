"""Convergence of the SURVEY 8d restart-stress leg (Davidson, max_dav=10, seed-2 guess) at several sizes/tolerances."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diaglib_amd import capi

ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
for n, t, m, tol in [(1_000_000, 32, 37, 1e-10), (10_000_000, 32, 37, 1e-9), (10_000_000, 32, 37, 1e-10), (10_000_000, 32, 37, 1e-11),
                     (10_000_000, 32, 37, 1e-12)]:
    ctx.synth_setup(n, 0, n)
    ev = ctx.panel(n, m); ctx.fill_guess(ev, 2)
    t0 = time.time()
    eig, _, ok, info = ctx.davidson_driver(n, t, m, 400, tol, 10, 0.0, mv, pc, ev)
    print(n, tol, ok, info, round(time.time() - t0, 2), eig[:3], flush=True)
    ev.free()
