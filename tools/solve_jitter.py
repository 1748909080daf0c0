"""Per-solve wall times of the headline configuration (Fortran driver through the C binding, as bench.py runs it): the spread over
solves of one process, with the driver's phase timers.   python tools/solve_jitter.py [solves] [knob6]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from diaglib_amd import capi

solves = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n, t, m = 2_000_000, 8, 13
c = capi.Context()
c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
if len(sys.argv) > 2:
    c.set_option(100 + 6, int(sys.argv[2]))
c.synth_setup(n, 0, n)
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
ev = c.panel(g)
ts = []
for r in range(solves + 3):
    ev.upload(g)
    c.sync()
    t0 = time.perf_counter()
    eig, _, ok, info = c.davidson_driver(n, t, m, 100, 2e-13, 20, 0.0, mv, pc, ev)
    c.sync()
    ts.append((time.perf_counter() - t0) * 1e3)
ts = np.array(ts[3:])
print("solves %d: min %.2f median %.2f mean %.2f max %.2f ms; iters %d" % (solves, ts.min(), np.median(ts), ts.mean(), ts.max(), info["iters"]))
print(" ".join("%.2f" % x for x in ts))
