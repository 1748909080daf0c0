#!/usr/bin/env python3
"""Randomised sweep of ortho_cd / ortho_vs_x (separate and contiguous panels, i.e. with and without the folded
pending-W sweep) against the oracle.    python tools/fuzz_ortho.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = np.finfo(np.float64).eps
ctx = capi.Context()
o = Oracle()
bad = 0
for it in range(cases):
    ctx.set_option(100 + 6, (0, 12, 13)[it % 3])      # sweep schedule of ortho_vs_x: shipped choice / five-sweep / three-pass always
    k = int(rng.choice([rng.integers(1, 17), rng.integers(17, 49)]))
    m = int(rng.choice([0, rng.integers(1, 40), rng.integers(40, 200)]))
    n = int(rng.integers(max(m + k, 8) * 2, max(m + k, 8) * 2 + 6000))
    if rng.random() < 0.6:
        n += n % 2
    x = np.linalg.qr(rng.standard_normal((n, max(m, 1))))[0][:, :m]
    mix = rng.standard_normal((m, k)) * rng.choice([0.0, 1.0, 30.0]) if m else 0
    scale = 10.0 ** rng.uniform(-6, 3, size=k)
    u = (rng.standard_normal((n, k)) + (x @ mix if m else 0)) * scale[None, :]
    x = np.asfortranarray(x); u = np.asfortranarray(u)
    want, _, st = o.ortho_vs_x(x if m else np.zeros((n, 0), order="F"), u) if m else (o.ortho_cd(u)[0], 0, 0)
    res = {}
    if m:
        contiguous = rng.random() < 0.6
        if contiguous:
            panel = ctx.panel(np.asfortranarray(np.hstack([x, u])))
            ctx.ortho_vs_x(panel.col(0, m), panel.col(m, k))
            got = panel.col(m, k).download()
            xa = panel.col(0, m).download()
            res["x_untouched"] = 0.0 if np.array_equal(xa, x) else 1e9
            panel.free()
        else:
            px, pu = ctx.panel(x), ctx.panel(u)
            ctx.ortho_vs_x(px, pu)
            got = pu.download()
            px.free(); pu.free()
        res["x_orth"] = np.abs(x.T @ got).max() / (50 * EPS)
    else:
        contiguous = False
        pu = ctx.panel(u)
        ctx.ortho_cd(pu)
        got = pu.download(); pu.free()
    res["orthonormal"] = np.abs(got.T @ got - np.eye(k)).max() / (50 * EPS)
    res["vs_oracle"] = np.abs(got - want).max() / 1e-9
    worst = max(res.values())
    if worst > 1.0 or not np.isfinite(worst) or st != 0:
        bad += 1
        print("FAIL", dict(n=n, m=m, k=k, contiguous=contiguous), {a: float(b) for a, b in res.items()}, flush=True)
print(f"{cases} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
