#!/usr/bin/env python3
"""A/B of the transpose-tile stride of the fused TRMM+Gram sweep (knob 6: 1 = old stride 16*KT+8) on ortho_cd
calls, one process.   python tools/tune_trmm.py [n] [rounds]"""
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
for k in (13, 16, 8):
    keep = ctx.panel(n, k); ctx.random_fill(keep)
    u = ctx.panel(n, k)
    res = {}
    for _ in range(rounds):
        for v in (1, 0):
            ctx.set_option(TUNE0 + 6, v)
            ctx.lib.dla_copy(ctx.h, u.ptr, keep.ptr, 8 * n * k)
            ctx.reset_stats()
            ctx.ortho_cd(u)
            for name, st in ctx.kernel_stats().items():
                if name.startswith("gemm_kernel<1, 2, 2, GemmArgsInl, true"):
                    res.setdefault(v, []).append((st["alg_bytes"] / st["ms"] / 1e6, st["ms"] / st["launches"] * 1e3))
    ctx.set_option(TUNE0 + 6, 0)
    print(f"k={k:3d}  " + "  ".join(f"{'old' if v else 'new'} stride: med {np.median([r[0] for r in res[v]]):7.1f} GB/s "
                                    f"{np.median([r[1] for r in res[v]]):6.1f} us" for v in sorted(res)), flush=True)
