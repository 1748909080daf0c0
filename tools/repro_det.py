"""Repeated identical solves on one context: the executed chain schedules (DIAGLIB_AMD_CHAIN_DEBUG=1) and the bits of the result."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from diaglib_amd import capi

n, t, m, sigma = 300_000, 8, 13, 0.5
c = capi.Context()
c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
if len(sys.argv) > 1:
    c.set_option(100 + 6, int(sys.argv[1]))
c.synth_setup(n, 0, n, 4, sigma)
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
outs = []
for r in range(3):
    print("== solve", r, flush=True)
    ev = c.panel(g)
    eig, _, ok, info = c.davidson_driver(n, t, m, 100, 1e-10, 20, 0.0, mv, pc, ev)
    outs.append(eig.copy())
    print("   iters", info["iters"], "bits equal to solve 0:", np.array_equal(eig, outs[0]), np.abs(eig - outs[0]).max(), flush=True)
