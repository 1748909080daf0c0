#!/usr/bin/env python3
"""A/B of the wide Ritz + P sweep (round-5 review, item 5): one wave per SIMD (ritz_kernel<4 | 5, ..., true>, tune knob 0 = 0)
against two wave groups per block, each with part of the column tiles (ritz_pair_kernel, knob 0 = 12 .. 15: pipeline depth
default / 2 / 1 / 0), interleaved in one process on the cfg 5 shape (n = 1e7 rows, 111 basis columns, 37 + 37 outputs) and on the
four-tile shape (30 + 30).  HIP-event time per launch and the HBM rate on the algorithmic bytes.
    python tools/ritz_pair_ab.py [n] [rounds]
Needs a library built with -DDLA_AB_VARIANTS (measured slower, profiles/r06/ritz_pair_ab.txt: not part of the product build)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = capi.Context()
rng = np.random.default_rng(0)
for l, m, k2 in ((111, 37, 37), (111, 30, 30)):
    pv, pav = ctx.panel(n, l), ctx.panel(n, l)
    ctx.fill_guess(pv, 3, 0); ctx.fill_guess(pav, 4, 0)
    y = np.asfortranarray(rng.standard_normal((l, m))); c2 = np.asfortranarray(rng.standard_normal((l, k2)))
    eig = rng.standard_normal(m); skip = np.zeros(m, np.int32)
    pe, pr, pa, pp, pap = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, k2), ctx.panel(n, k2)
    # the two-group kernels must give the one-group kernel's bits (same contraction order per output element)
    outs = {}
    for knob in (0, 12, 13, 14, 15):
        ctx.set_option(100 + 0, knob); ctx.reset_stats()
        rn = ctx.ritz_residual_p(pv, pav, y, eig, m, skip, pe, pr, pa, c2, pp, pap)
        names = [nm for nm in ctx.kernel_stats() if nm.startswith("ritz") and "reduce" not in nm]
        if knob and not any(nm.startswith("ritz_pair_kernel") for nm in names):
            raise SystemExit("this library was built without -DDLA_AB_VARIANTS: the two-group kernels are not in it")
        outs[knob] = (rn.copy(), pr.col(0, 1).download(), pp.col(k2 - 1, 1).download(), pap.col(0, 1).download())
    for knob in (12, 13, 14, 15):
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[knob])), f"knob {knob}: results differ from the one-group kernel"
    print(f"l={l} m={m}+{k2}: all four two-group variants bit-identical to the one-group kernel (norms, first / last columns of r, P, AP)", flush=True)
    res = {}
    for r in range(rounds):
        for knob in (0, 12, 13, 14, 15):
            ctx.set_option(100 + 0, knob)
            ctx.ritz_residual_p(pv, pav, y, eig, m, skip, pe, pr, pa, c2, pp, pap)       # warm-up
            ctx.reset_stats(); ctx.set_option(capi.OPT_PROFILE, 1)
            for _ in range(4):
                ctx.ritz_residual_p(pv, pav, y, eig, m, skip, pe, pr, pa, c2, pp, pap)
            ks = ctx.kernel_stats(); ctx.set_option(capi.OPT_PROFILE, 0)
            for name, v in ks.items():
                if name.startswith("ritz") and "reduce" not in name:
                    res.setdefault((knob, name), []).append((v["ms"] / v["launches"], v["alg_bytes"] / v["ms"] / 1e6))
    ctx.set_option(100 + 0, 0)
    for (knob, name), v in sorted(res.items()):
        print(f"l={l} m={m}+{k2} knob0={knob:2d} {name:48s} " + "  ".join(f"{ms:7.3f} ms {gb:6.0f} GB/s" for ms, gb in v), flush=True)
    for p_ in (pv, pav, pe, pr, pa, pp, pap):
        p_.free()
