#!/usr/bin/env python3
"""A/B of one engine knob on the fused sweeps of real ortho_vs_x calls, one process.
knob 7: column-step pipeline depth of the fused projection sweep (0 = none, 4, 8);
knob 3: cap on resident 4-wave blocks per CU of the panel-product kernels (0 = 4, else up to the LDS limit).
    python tools/tune_fused.py [n] [rounds] [knob] [v1,v2,...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
knob = int(sys.argv[3]) if len(sys.argv) > 3 else 7
values = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0, 4, 8]
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
TUNE0 = 100
k = 13
for L in (26, 65, 117):
    panel = ctx.panel(n, L + k)
    ctx.random_fill(panel)
    for j in range(0, L, 13):                      # orthonormal X, block by block
        if j:
            ctx.ortho_vs_x(panel.col(0, j), panel.col(j, min(13, L - j)))
        else:
            ctx.ortho_cd(panel.col(0, 13))
    keep = ctx.panel(n, k); ctx.random_fill(keep)
    res = {}
    outs = {}
    for _ in range(rounds):
        for v in values:
            ctx.set_option(TUNE0 + knob, v)
            ctx.lib.dla_copy(ctx.h, panel.col(L, k).ptr, keep.ptr, 8 * n * k)
            ctx.reset_stats()
            ctx.ortho_vs_x(panel.col(0, L), panel.col(L, k))
            outs[v] = panel.col(L, k).download()
            for name, st in ctx.kernel_stats().items():
                if name.startswith(("gemm_kernel<1, 2, 0, GemmArgs, true", "gemm_kernel<1, 2, 1, GemmArgs, true",
                                    "gemm_lds_kernel<1, 0, true", "gemm_lds_kernel<1, 1, true")):
                    res.setdefault(("proj", v), []).append(st["alg_bytes"] / st["ms"] / 1e6)
                if name.startswith("gemm_kernel<1, 2, 2, GemmArgsInl, true"):
                    res.setdefault(("trmm", v), []).append(st["alg_bytes"] / st["ms"] / 1e6)
    ctx.set_option(TUNE0 + knob, 0)
    print(f"L={L:4d} max |U(v) - U(v0)| =", [float(np.abs(outs[v] - outs[values[0]]).max()) for v in values])
    for what in ("proj", "trmm"):
        print(f"L={L:4d} {what}  " + "  ".join(f"knob{knob}={v}: med {np.median(res[(what, v)]):7.1f} GB/s" for v in values), flush=True)
