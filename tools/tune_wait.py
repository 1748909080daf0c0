#!/usr/bin/env python3
"""A/B of the host wait primitive (knob 6: 0 = event record + query, 2 = stream query) on whole solves, interleaved
in one process.   python tools/tune_wait.py [n] [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
t, m = 8, 13
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
gd = ctx.panel(g); ev = ctx.panel(n, m)
res = {0: [], 2: []}
for r in range(rounds + 1):
    for v in (0, 2):
        ctx.set_option(100 + 6, v)
        ctx.lib.dla_copy(ctx.h, ev.ptr, gd.ptr, 8 * n * m)
        ctx.sync()
        t0 = time.perf_counter()
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        dt = time.perf_counter() - t0
        if r:
            res[v].append(dt * 1e3)
ctx.set_option(100 + 6, 0)
for v in res:
    print(f"knob6={v}: median {np.median(res[v]):7.3f} ms  min {min(res[v]):7.3f} ms  ({info['iters']} iterations)")
