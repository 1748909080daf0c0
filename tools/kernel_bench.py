#!/usr/bin/env python3
"""Per-shape bandwidth of the block-algebra kernels (HIP-event time from the engine's own
statistics, algorithmic bytes as in DESIGN.md section 3).  Tuning aid:

    python tools/kernel_bench.py [n] [reps]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
lmax = 260
big = ctx.panel(n, lmax + 16); ctx.random_fill(big)
big2 = ctx.panel(n, lmax); ctx.random_fill(big2)
out1 = ctx.panel(n, 16); out2 = ctx.panel(n, 16); out3 = ctx.panel(n, 16)
rng = np.random.default_rng(0)


def run(name, cls, f):
    f(); ctx.reset_stats()
    for _ in range(reps):
        f()
    st = ctx.stats()[cls]
    print(f"{name:34s} {st['ms'] / reps * 1e3:9.1f} us  {st['alg_bytes'] / st['ms'] / 1e6:8.1f} GB/s  "
          f"({st['launches'] // reps} launches)", flush=True)


for k in (13,):
    u = big.col(lmax, k)
    run(f"gram self k={k}", "gram", lambda: ctx.gram(u, u))
    for l in (13, 26, 39, 65, 104, 130, 195, 260):
        x = big.col(0, l)
        c = np.asfortranarray(rng.standard_normal((l, k)) * 1e-3)
        y = np.asfortranarray(rng.standard_normal((l, k)))
        run(f"gram   L={l:3d} k={k}", "gram", lambda: ctx.gram(x, u))
        run(f"update L={l:3d} k={k}", "gemm", lambda: ctx.panel_update(x, c, out1.col(0, k)))
        run(f"ritz   L={l:3d} M={k}", "ritz", lambda: ctx.ritz_residual(x, big2.col(0, l), y, np.ones(k), 8, np.zeros(k, np.int32),
                                                                 out2.col(0, k), out3.col(0, k)))
    w = np.asfortranarray(np.tril(rng.standard_normal((k, k))) + 3 * np.eye(k))
    run(f"trmm k={k}", "trmm", lambda: ctx.trmm_linvt(out1.col(0, k), w))
