#!/usr/bin/env python3
"""Interleaved A/B of the instruction-scheduling variants of the wide Ritz + P sweep (ritz_kernel<4 / 5, ..., true, SCHED>):
tune knob 0 = 0 (compiler's schedule), 7 / 9 (__builtin_amdgcn_iglp_opt(0) / (1)), 8 / 10 (two explicit sched_group_barrier
pipelines; 9 and 10 exist for five tiles only), 11 (two stages per loop trip, the register sets changing roles).
Checks that the three variants give the same bits, then times them in one process on one device.
Needs a library built with -DDLA_AB_VARIANTS (the variants are not part of the product build since round 5).

    python tools/ritz_sched_ab.py [n] [rounds]
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
rng = np.random.default_rng(0)

for (l, m) in ((111, 37), (96, 32)):
    KNOBS = (0, 7, 8, 9, 10, 11) if m == 37 else (0, 7, 8, 11)
    v = ctx.panel(n, l); ctx.random_fill(v)
    av = ctx.panel(n, l); ctx.random_fill(av)
    ev, r, avy, p, ap = (ctx.panel(n, m) for _ in range(5))
    y = np.asfortranarray(rng.standard_normal((l, m)))
    c2 = np.asfortranarray(rng.standard_normal((l, m)))
    eig = np.linspace(1.0, 2.0, m); skip = np.zeros(m, np.int32)

    def run():
        return ctx.ritz_residual_p(v, av, y, eig, m, skip, ev, r, avy, c2, p, ap)

    ref = None
    for knob in KNOBS:
        ctx.set_option(TUNE0, knob)
        rn = run()
        got = (rn.copy(), r.download()[:4096].copy(), ap.download()[:4096].copy())
        if ref is None:
            ref = got
        else:
            same = all(np.array_equal(a, b) for a, b in zip(ref, got))
            print(f"L={l} M={m}+{m}: knob {knob} bits equal to knob 0: {same}", flush=True)
    res = {k: [] for k in KNOBS}
    for _ in range(rounds):
        for knob in KNOBS:
            ctx.set_option(TUNE0, knob)
            run(); ctx.reset_stats()
            for _ in range(4):
                run()
            st = ctx.stats()["ritz"]
            res[knob].append((st["ms"] / 4 * 1e3, st["alg_bytes"] / st["ms"] / 1e6))
    ctx.set_option(TUNE0, 0)
    for knob in KNOBS:
        us = np.median([a for a, _ in res[knob]]); gb = np.median([b for _, b in res[knob]])
        print(f"L={l} M={m}+{m} knob {knob}: median {us:8.1f} us  {gb:7.1f} GB/s", flush=True)
    for q in (v, av, ev, r, avy, p, ap):
        q.free()
