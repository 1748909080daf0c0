#!/usr/bin/env python3
"""A/B of the Gram grid size (knob 4: blocks per pass; 0 = one per CU) for narrow passes, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from diaglib_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = capi.Context(); ctx.set_option(capi.OPT_PROFILE, 1)
big = ctx.panel(n, 64); ctx.random_fill(big)
u13 = ctx.panel(n, 13); ctx.random_fill(u13)
def rate(f, reps=8):
    f(); ctx.reset_stats()
    for _ in range(reps): f()
    ks = ctx.kernel_stats()
    main = {k: v for k, v in ks.items() if k.startswith("gram_") and "reduce" not in k and v["ms"] > 0}
    k = max(main, key=lambda q: main[q]["ms"]); v = main[k]
    return k, v["alg_bytes"] / v["ms"] / 1e6
for (l, uu) in ((4, u13), (13, None), (13, u13), (26, u13), (39, u13)):
    x = big.col(0, l); u = x if uu is None else uu
    res = {}
    for _ in range(rounds):
        for v in (0, 512, 768, 1024):
            ctx.set_option(104, v)
            name, r = rate(lambda: ctx.gram(x, u))
            res.setdefault(v, []).append(r)
    ctx.set_option(104, 0)
    print(f"L={l:3d} k={u.m:3d} {name:30s} " + "  ".join(f"{v}: {np.median(res[v]):7.1f}" for v in res), flush=True)
