#!/usr/bin/env python3
"""Whole-solve A/B of one engine knob on the wide-block configurations (BASELINE configs[3]/[4] shapes), interleaved in
one process: wall time per solve and the rate of every kernel class.

    python tools/wide_solve_ab.py <knob> <v1,v2,...> [lobpcg|davidson] [n] [roots] [n_max] [rounds]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

knob = int(sys.argv[1]); values = [int(v) for v in sys.argv[2].split(",")]
solver = sys.argv[3] if len(sys.argv) > 3 else "lobpcg"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 10_000_000
t = int(sys.argv[5]) if len(sys.argv) > 5 else 32
m = int(sys.argv[6]) if len(sys.argv) > 6 else 37
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 2
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
gd = ctx.panel(g); del g
ev = ctx.panel(n, m)


def solve():
    ctx.lib.dla_copy(ctx.h, ev.ptr, gd.ptr, 8 * n * m); ctx.sync()
    t0 = time.perf_counter()
    if solver == "lobpcg":
        eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 200, 2e-13, 0.0, mv, pc, ev)
    else:
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
    return (time.perf_counter() - t0) * 1e3, ok, info


wall = {v: [] for v in values}; cls = {v: {} for v in values}
for r in range(rounds + 1):
    for v in values:
        ctx.set_option(100 + knob, v)
        ctx.set_option(capi.OPT_PROFILE, 0)
        dt, ok, info = solve()
        ctx.set_option(capi.OPT_PROFILE, 1); ctx.reset_stats()
        solve()
        st = ctx.stats()
        last_kernels = ctx.kernel_stats()
        if r:
            wall[v].append(dt)
            for c, s in st.items():
                if isinstance(s, dict) and s.get("ms", 0) > 0:
                    cls[v].setdefault(c, []).append((s["alg_bytes"] / s["ms"] / 1e6, s["ms"]))
ctx.set_option(100 + knob, 0)
print(f"{solver} n={n} roots={t} n_max={m}: {info['iters']} iterations, converged={ok}")
for v in values:
    line = "  ".join(f"{c} {np.median([a for a, _ in x]):6.0f} GB/s {np.median([b for _, b in x]):6.1f} ms" for c, x in cls[v].items())
    print(f"knob{knob}={v}: wall median {np.median(wall[v]):8.2f} ms   {line}", flush=True)
print("kernels of the last solve (knob value %d):" % values[-1])
for k, v in sorted(last_kernels.items(), key=lambda kv: -kv[1]["ms"]):
    if v["ms"] > 0:
        print(f"  {k:62s} {v['launches']:4d} x {v['ms'] / max(v['launches'], 1) * 1e3:9.1f} us  {v['alg_bytes'] / v['ms'] / 1e6:7.0f} GB/s  {v['ms']:7.1f} ms")
