#!/usr/bin/env python3
"""Interleaved A/B of kernel-shape knobs in ONE process on ONE device (cdna_hip_programming.md rule 24:
timings from different boxes differ by ~10 %, so variants are only ever ranked inside one run).

    python tools/tune_ab.py [n] [rounds]
"""
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
lmax = 130
big = ctx.panel(n, lmax + 16); ctx.random_fill(big)
big2 = ctx.panel(n, lmax); ctx.random_fill(big2)
o1 = ctx.panel(n, 16); o2 = ctx.panel(n, 16); o3 = ctx.panel(n, 16)
rng = np.random.default_rng(0)
k = 13


def timeit(cls, f, reps=6):
    f(); ctx.reset_stats()
    for _ in range(reps):
        f()
    st = ctx.stats()[cls]
    return st["ms"] / reps * 1e3, st["alg_bytes"] / st["ms"] / 1e6


def ab(title, cls, f, knob, values):
    res = {v: [] for v in values}
    for _ in range(rounds):
        for v in values:
            ctx.set_option(TUNE0 + knob, v)
            res[v].append(timeit(cls, f)[1])
    ctx.set_option(TUNE0 + knob, 0)
    line = "  ".join(f"{v}: med {np.median(res[v]):7.1f} max {max(res[v]):7.1f}" for v in values)
    print(f"{title:28s} knob{knob}  {line}", flush=True)


for (l, kw) in ((100, 21), (111, 37)):
    x = big.col(0, l); x2 = big2.col(0, l)
    ow1 = ctx.panel(n, kw); ow2 = ctx.panel(n, kw); ow3 = ctx.panel(n, kw)
    y = np.asfortranarray(rng.standard_normal((l, kw)))
    c = np.asfortranarray(rng.standard_normal((l, kw)) * 1e-3)
    eig = np.ones(kw); skip = np.zeros(kw, np.int32)
    ab(f"ritz L={l} M={kw} pipeline", "ritz", lambda: ctx.ritz_residual(x, x2, y, eig, kw, skip, ow2, ow3), 0, [1, 0, 4])
    ab(f"gemm L={l} k={kw} pipeline", "gemm", lambda: ctx.panel_gemm(x, c, ow1), 2, [1, 0, 4])
    ab(f"update L={l} k={kw} pipeline", "gemm", lambda: ctx.panel_update(x, c, ow1), 2, [1, 0, 4])
    ab(f"gram L={l} k={kw} prefetch", "gram", lambda: ctx.gram(x, ow1), 5, [0, 1])
    ab(f"gram {kw}x{kw} self prefetch", "gram", lambda: ctx.gram(ow1, ow1), 5, [0, 1])
    ab(f"gram L={l} x L (S^T AS) prefetch", "gram", lambda: ctx.gram(x, x2), 5, [0, 1])
