#!/usr/bin/env python3
"""Back-to-back rate of the fused Ritz sweep and the projection sweep (no host gaps, no other kernels in between),
for comparison with their in-solve rates.   python tools/ritz_b2b.py [n]"""
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
M = 13
v = ctx.panel(n, 130); ctx.random_fill(v)
av = ctx.panel(n, 130); ctx.random_fill(av)
e = ctx.panel(n, M); r = ctx.panel(n, M); z = ctx.panel(n, M)
for L in (26, 65, 117):
    y = np.asfortranarray(rng.standard_normal((L, M)))
    c = np.asfortranarray(rng.standard_normal((L, M)) * 1e-3)
    eig = np.ones(M); skip = np.zeros(M, np.int32)
    for name, f in (("ritz", lambda: ctx.ritz_residual(v.col(0, L), av.col(0, L), y, eig, 8, skip, e, r)),
                    ("gemm", lambda: ctx.panel_gemm(v.col(0, L), c, z)),
                    ("update", lambda: ctx.panel_update(v.col(0, L), c, z)),
                    ("gram", lambda: ctx.gram(v.col(0, L), z))):
        f(); ctx.reset_stats()
        for _ in range(8):
            f()
        ks = ctx.kernel_stats()
        main = {k: s for k, s in ks.items() if s["alg_bytes"] > 0 and s["ms"] > 0}
        k = max(main, key=lambda q: main[q]["ms"])
        print(f"L={L:4d} {name:7s} {k:48s} {main[k]['ms'] / main[k]['launches'] * 1e3:7.1f} us  {main[k]['alg_bytes'] / main[k]['ms'] / 1e6:7.1f} GB/s", flush=True)
