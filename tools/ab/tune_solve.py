#!/usr/bin/env python3
"""A/B of one engine knob on whole benchmark solves, interleaved in one process: wall time per solve and the rate of
one kernel family.   python tools/tune_solve.py <knob> <v1,v2,...> [kernel-prefix] [n] [rounds]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

knob = int(sys.argv[1]); values = [int(v) for v in sys.argv[2].split(",")]
prefix = sys.argv[3] if len(sys.argv) > 3 else "ritz_kernel"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 2_000_000
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 6
t, m = 8, 13
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
gd = ctx.panel(g); ev = ctx.panel(n, m)
wall = {v: [] for v in values}; rate = {v: [] for v in values}
for r in range(rounds + 1):
    for v in values:
        ctx.set_option(100 + knob, v)
        ctx.lib.dla_copy(ctx.h, ev.ptr, gd.ptr, 8 * n * m); ctx.sync()
        ctx.set_option(capi.OPT_PROFILE, 0)
        t0 = time.perf_counter()
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        dt = time.perf_counter() - t0
        ctx.lib.dla_copy(ctx.h, ev.ptr, gd.ptr, 8 * n * m)
        ctx.set_option(capi.OPT_PROFILE, 1); ctx.reset_stats()
        ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        ks = ctx.kernel_stats()
        by = sum(s["alg_bytes"] for k, s in ks.items() if k.startswith(prefix)); ms = sum(s["ms"] for k, s in ks.items() if k.startswith(prefix))
        if r:
            wall[v].append(dt * 1e3); rate[v].append(by / max(ms, 1e-9) / 1e6)
ctx.set_option(100 + knob, 0)
for v in values:
    print(f"knob{knob}={v}: wall median {np.median(wall[v]):7.3f} ms  {prefix} {np.median(rate[v]):7.1f} GB/s  ({info['iters']} iterations)")
