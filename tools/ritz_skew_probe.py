#!/usr/bin/env python3
"""Does the rate of the fused Ritz sweep depend on where its two input panels lie relative to each other?  The sweep is the only
kernel of a solve that streams TWO panels at the same column / row offsets at once (V and A V); its time differs by 10-15 %
between boxes of the pool while every other sweep's is the same.  Both panels are carved out of ONE allocation here, A V at
V + 130 columns + a byte offset, and the sweep is timed back to back for a range of offsets (16-byte aligned).
    python tools/ritz_skew_probe.py [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
rng = np.random.default_rng(0)
M, L = 13, 117
col = 8 * n
arena = ctx.panel(n, 2 * 130 + 40); ctx.random_fill(arena)
e = ctx.panel(n, M); r = ctx.panel(n, M)
y = np.asfortranarray(rng.standard_normal((L, M)))
eig = np.ones(M); skip = np.zeros(M, np.int32)
print(f"arena at {arena.ptr:#x}, e at {e.ptr:#x}, r at {r.ptr:#x}")
offsets = [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 1 << 21, (1 << 21) + 65536, 3 << 20, 1 << 22, 1 << 23,
           (1 << 23) + (1 << 12), 1 << 24, 1 << 25, (1 << 25) + 777 * 16]
for rep in range(2):
    for off in offsets:
        v = capi.DevPanel(ctx, n, L, arena.ptr, owner=False)
        av = capi.DevPanel(ctx, n, L, arena.ptr + 130 * col + off, owner=False)
        ctx.ritz_residual(v, av, y, eig, 8, skip, e, r)
        ctx.reset_stats()
        for _ in range(6):
            ctx.ritz_residual(v, av, y, eig, 8, skip, e, r)
        ks = ctx.kernel_stats()
        k = [q for q in ks if q.startswith("ritz_kernel")][0]
        us = ks[k]["ms"] / ks[k]["launches"] * 1e3
        print(f"rep {rep} offset {off:>10d} B (delta mod 2 MiB = {(130 * col + off) % (1 << 21):>8d}): {us:7.1f} us  {ks[k]['alg_bytes'] / ks[k]['ms'] / 1e6:7.1f} GB/s", flush=True)
