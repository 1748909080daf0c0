#!/usr/bin/env python3
"""Timing of the projection sweep that measures what it stores (OP_COMBOX, gram_lds_kernel<..., 2>) inside ortho_vs_x chains, per basis
width, for several blocks-per-pass settings (tune knob 4) -- interleaved, one process.   python tools/combox_ab.py [n] [rounds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "512"])]
ctx = capi.Context()
ctx.set_option(capi.OPT_PROFILE, 1)
TUNE0 = 100
rng = np.random.default_rng(5)
k = 13
mmax = 104
big = ctx.panel(n, mmax + k)
ctx.random_fill(big)
# orthonormal X block by block on the device
ctx.ortho_cd(big.col(0, k))
for c0 in range(k, mmax, k):
    ctx.ortho_vs_x(big.col(0, c0), big.col(c0, k))
for m in (26, 52, 65, 78, 104):
    res = {v: [] for v in variants}
    name = None
    for _ in range(rounds):
        for v in variants:
            ctx.set_option(TUNE0 + 4, v)
            ctx.random_fill(big.col(m, k))
            ctx.ortho_vs_x(big.col(0, m), big.col(m, k))          # (plan)
            ctx.reset_stats()
            for _ in range(3):
                ctx.random_fill(big.col(m, k))
                ctx.ortho_vs_x(big.col(0, m), big.col(m, k))
            ks = ctx.kernel_stats()
            for kn, st in ks.items():
                if kn.startswith("gram_lds_kernel") and kn.endswith(", 2>") and st["ms"] > 0:
                    res[v].append(st["alg_bytes"] / st["ms"] / 1e6); name = kn
    ctx.set_option(TUNE0 + 4, 0)
    print(f"m={m:4d} {name}: " + "  ".join(f"knob4={v}: {np.median(res[v]):7.1f} GB/s" for v in variants), flush=True)
