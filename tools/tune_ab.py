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


u13 = big.col(lmax, k)
ab("gram self k=13 blocks/pass", "gram", lambda: ctx.gram(u13, u13), 4, [256, 512, 768, 1024, 2048])
ab("gram L=13 blocks/pass", "gram", lambda: ctx.gram(big.col(0, 13), u13), 4, [256, 512, 768, 1024, 2048])
ab("gram L=26 blocks/pass", "gram", lambda: ctx.gram(big.col(0, 26), u13), 4, [256, 512, 768, 1024])
ab("gram L=39 blocks/pass", "gram", lambda: ctx.gram(big.col(0, 39), u13), 4, [256, 512, 768, 1024])
ab("gram L=52 blocks/pass", "gram", lambda: ctx.gram(big.col(0, 52), u13), 4, [256, 512, 768])
ab("gram L=65 blocks/pass", "gram", lambda: ctx.gram(big.col(0, 65), u13), 4, [256, 512, 768])
