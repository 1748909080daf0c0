#!/usr/bin/env python3
"""A/B of the two Gram kernels (knob 5: 2 = fragment loads straight from HBM, 0 = default: full-line loads staged
through LDS where instantiated) in one process, with a result comparison.   python tools/tune_gram.py [n] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
TUNE0 = 100
big = ctx.panel(n, 264); ctx.random_fill(big)
u13 = ctx.panel(n, 13); ctx.random_fill(u13)
u21 = ctx.panel(n, 21); ctx.random_fill(u21)
u37 = ctx.panel(n, 37); ctx.random_fill(u37)


def kernel_time(f, reps=6):
    f(); ctx.reset_stats()
    for _ in range(reps):
        f()
    ks = ctx.kernel_stats()
    main = {k: v for k, v in ks.items() if k.startswith("gram_") and "reduce" not in k and v["ms"] > 0}
    name = max(main, key=lambda k: main[k]["ms"])
    v = main[name]
    return name, v["ms"] / v["launches"] * 1e3, v["alg_bytes"] / v["ms"] / 1e6


for (l, u) in ((4, u13), (13, u13), (26, u13), (39, u13), (52, u13), (78, u13), (104, u13), (143, u13), (156, u13), (247, u13), (65, u13), (91, u13), (117, u13), (130, u13), (180, u13), (100, u21), (111, u37), (13, None)):
    x = big.col(0, l)
    uu = x if u is None else u
    res = {2: [], 0: []}
    names = {}
    outs = {}
    for _ in range(rounds):
        for v in (2, 0):
            ctx.set_option(TUNE0 + 5, v)
            nm, us, gbs = kernel_time(lambda: ctx.gram(x, uu))
            res[v].append(gbs); names[v] = nm
            outs[v] = ctx.gram(x, uu)
    ctx.set_option(TUNE0 + 5, 0)
    err = np.abs(outs[0] - outs[2]).max() / max(1e-300, np.abs(outs[2]).max())
    m = {v: np.median(res[v]) for v in res}
    print(f"L={l:4d} k={uu.m:3d}  {names[2]:30s} {m[2]:7.1f} | {names[0]:28s} {m[0]:7.1f} ({m[0] / m[2]:.3f})  rel diff {err:.1e}", flush=True)
