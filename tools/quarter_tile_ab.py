#!/usr/bin/env python3
"""Interleaved A/B of the wide-block kernel variants (knob 7, see ab()) at the wide-block shapes of
BASELINE configs[3]/[4] (n_max = 21 and 37), one process, one device.

    python tools/quarter_tile_ab.py [n] [rounds]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
TUNE0 = 100
rng = np.random.default_rng(0)


def timeit(cls, f, reps=4):
    f(); ctx.reset_stats()
    for _ in range(reps):
        f()
    st = ctx.stats()[cls]
    return st["alg_bytes"] / st["ms"] / 1e6


LABEL = {0: "default", 1: "full tiles", 2: "pads loaded", 3: "narrow passes", 4: "no next-tile prefetch", 5: "64-row tiles", 8: "multi-pass lower"}


def ab(title, cls, f, values=(1, 0), knob=7, labels=None):
    # knob 7: 1 = full 16x16x4 tiles only, 2 = Gram loads its padded / unused column groups too, 3 = Gram passes of at most
    # 12 accumulator tiles, 4 = no next-tile prefetch in the row products, 5 = 64-row wave tiles in the fused three-tile
    # sweeps, 8 = lower triangle of two panels in several passes, 0 = default
    res = {v: [] for v in values}
    for _ in range(rounds):
        for v in values:
            ctx.set_option(TUNE0 + knob, v)
            res[v].append(timeit(cls, f))
    ctx.set_option(TUNE0 + knob, 0)
    base = np.median(res[values[0]])
    lab = labels or LABEL
    print(f"{title:34s} " + "   ".join(f"{lab[v]} {np.median(res[v]):7.1f}" for v in values) +
          f" GB/s   ({np.median(res[0]) / base - 1:+.1%})", flush=True)


for (l, kw) in ((111, 37), (63, 21)):
    big = ctx.panel(n, l + kw); ctx.random_fill(big)
    x = big.col(0, l); blk = big.col(l, kw)
    x2 = ctx.panel(n, l); ctx.random_fill(x2)
    ow1 = ctx.panel(n, kw); ow2 = ctx.panel(n, kw)
    y = np.asfortranarray(rng.standard_normal((l, kw)) / np.sqrt(l))
    c = np.asfortranarray(rng.standard_normal((l, kw)) * 1e-3)
    w = np.asfortranarray(np.triu(rng.standard_normal((kw, kw)) * 1e-2) + np.eye(kw))
    cp = np.asfortranarray(np.vstack([-c, w]))
    eig = np.ones(kw); skip = np.zeros(kw, np.int32)
    ab(f"gram self {kw}x{kw}", "gram", lambda: ctx.gram(blk, blk), (1, 2, 0))
    ab(f"gram L={l} x {kw}", "gram", lambda: ctx.gram(x, blk), (1, 2, 3, 0))
    ab(f"gram L={2 * kw} x {kw}", "gram", lambda: ctx.gram(x.col(0, 2 * kw), blk), (1, 2, 3, 0))
    ab(f"gram lower {l} x {l} (S^T AS)", "gram", lambda: ctx.gram_lower(x, x2), (8, 0))
    ab(f"gram lower {l - 22} x {l - 22}", "gram", lambda: ctx.gram_lower(x.col(0, l - 22), x2.col(0, l - 22)), (8, 0))
    ab(f"gemm L={l} k={kw}", "gemm", lambda: ctx.panel_gemm(x, c, ow1))
    ab(f"update L={l} k={kw}", "gemm", lambda: ctx.panel_update(x, c, ow1))
    ab(f"trmm k={kw}", "trmm", lambda: ctx.trmm_linvt(blk, np.asfortranarray(w.T)))
    ab(f"trmm+gram k={kw}", "trmm", lambda: ctx.trmm_gram(blk, w), (1, 4, 5, 0))
    ab(f"update+gram L={l} k={kw}", "gemm", lambda: ctx.update_gram(x, c, blk), (1, 4, 5, 0))
    ab(f"[X|U] sweep+gram m={l} k={kw}", "gemm", lambda: ctx.combo_gram(x, cp, blk), (1, 4, 5, 0))
    ab(f"ritz L={l} M={kw}", "ritz", lambda: ctx.ritz_residual(x, x2, y, eig, kw, skip, ow1, ow2))
    for p in (big, x2, ow1, ow2):
        p.free()
