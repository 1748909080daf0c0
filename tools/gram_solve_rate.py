#!/usr/bin/env python3
"""Rates of the Gram kernels inside benchmark solves (HIP-event time per kernel name), for comparing builds or knobs on
one box:   python tools/gram_solve_rate.py [n] [knob7 values, e.g. 0,2]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
values = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
t, m = 8, 13
ctx = capi.Context(); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1); ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
gd = ctx.panel(g); ev = ctx.panel(n, m)
acc = {v: {} for v in values}
for rep in range(5):
    for v in values:
        ctx.set_option(107, v)
        ctx.lib.dla_copy(ctx.h, ev.ptr, gd.ptr, 8 * n * m); ctx.sync()
        ctx.set_option(capi.OPT_PROFILE, 1); ctx.reset_stats()
        ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        if rep:
            for k, s in ctx.kernel_stats().items():
                if s["ms"] > 0:
                    acc[v].setdefault(k, []).append(s["ms"] / s["launches"] * 1e3)
ctx.set_option(107, 0)
for k in sorted(acc[values[0]]):
    print(f"{k:58s} " + "  ".join(f"knob{v}: {np.median(acc[v].get(k, [0])):8.1f} us" for v in values))
