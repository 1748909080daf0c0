#!/usr/bin/env python3
"""Probe: is the row-contiguous (gemm-layout) read path or the write traffic what holds the panel
products at ~5 TB/s?  Z = X C with k = 1 (almost read-only) against k = 13, and the Gram at the same L."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from diaglib_amd import capi
n = 2_000_000
ctx = capi.Context(); ctx.set_option(capi.OPT_PROFILE, 1)
big = ctx.panel(n, 146); ctx.random_fill(big)
out = ctx.panel(n, 16)
rng = np.random.default_rng(0)
def t(cls, f, reps=8):
    f(); ctx.reset_stats()
    for _ in range(reps): f()
    st = ctx.stats()[cls]; return st["alg_bytes"] / st["ms"] / 1e6, st["ms"] / reps * 1e3
for l in (39, 117):
    x = big.col(0, l)
    for k in (1, 4, 13):
        c = np.asfortranarray(rng.standard_normal((l, k)))
        print(f"gemm Z=XC  L={l:3d} k={k:2d}: %.1f GB/s  %.1f us" % t("gemm", lambda: ctx.panel_gemm(x, c, out.col(0, k))), flush=True)
        print(f"gram X^T U L={l:3d} k={k:2d}: %.1f GB/s  %.1f us" % t("gram", lambda: ctx.gram(x, big.col(130, k))), flush=True)
