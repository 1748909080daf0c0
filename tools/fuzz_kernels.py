#!/usr/bin/env python3
"""Randomised shape sweep of the block kernels against the oracle (float64 dot-product bound).
    python tools/fuzz_kernels.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = np.finfo(np.float64).eps
ctx = capi.Context()
o = Oracle()
bad = 0
for it in range(cases):
    n = int(rng.choice([rng.integers(1, 70), rng.integers(70, 700), rng.integers(700, 9000)]))
    if rng.random() < 0.6:
        n += n % 2                                  # mostly even n (16-byte paths)
    l = int(rng.choice([rng.integers(1, 17), rng.integers(17, 130), rng.integers(130, 300)]))
    k = int(rng.choice([rng.integers(1, 17), rng.integers(17, 49)]))
    x = np.asfortranarray(rng.standard_normal((n, l)))
    u = np.asfortranarray(rng.standard_normal((n, k)))
    c = np.asfortranarray(rng.standard_normal((l, k)))
    px, pu = ctx.panel(x), ctx.panel(u)
    errs = {}
    got = ctx.gram(px, pu)
    errs["gram"] = np.max(np.abs(got - o.gemm_tn(x, u)) / (64 * EPS * (np.abs(x).T @ np.abs(u)) + 1e-300))
    if l <= 300:
        gl = ctx.gram_lower(px, ctx.panel(x[:, ::-1].copy(order="F")))
        want = o.gemm_tn(x, np.asfortranarray(x[:, ::-1]))
        low = np.tril(np.ones((l, l), bool))
        errs["gram_lower"] = np.max((np.abs(gl - want) / (64 * EPS * (np.abs(x).T @ np.abs(x[:, ::-1])) + 1e-300))[low])
    pz = ctx.panel(n, k)
    ctx.panel_gemm(px, c, pz)
    bound = 64 * EPS * (np.abs(x) @ np.abs(c)) + 1e-300
    errs["gemm"] = np.max(np.abs(pz.download() - o.gemm_nn(x, c)) / bound)
    ctx.panel_update(px, c, pu)
    errs["update"] = np.max(np.abs(pu.download() - o.gemm_nn(x, c, alpha=-1.0, beta=1.0, z=u)) / (bound + 4 * EPS * np.abs(u)))
    m = min(k, 48)
    y = np.asfortranarray(rng.standard_normal((l, m)))
    av = np.asfortranarray(rng.standard_normal((n, l)))
    eig = rng.standard_normal(m)
    nres = int(rng.integers(0, m + 1))
    skip = (rng.random(m) < 0.3).astype(np.int32)
    pe, pr = ctx.panel(n, m), ctx.panel(n, m)
    rn = ctx.ritz_residual(px, ctx.panel(av), y, eig, nres, skip, pe, pr)
    ev = x @ y
    r = av @ y
    for i in range(nres):
        if not skip[i]:
            r[:, i] -= eig[i] * ev[:, i]
    b2 = 64 * EPS * (np.abs(x) @ np.abs(y) * (1 + np.abs(eig)[None, :]) + np.abs(av) @ np.abs(y)) + 1e-300
    errs["ritz_evec"] = np.max(np.abs(pe.download() - ev) / b2)
    errs["ritz_r"] = np.max(np.abs(pr.download() - r) / b2)
    worst = max(errs.values())
    if worst > 1.0 or not np.isfinite(worst):
        bad += 1
        print("FAIL", dict(n=n, l=l, k=k, nres=nres), {a: float(b) for a, b in errs.items()}, flush=True)
    for p in (px, pu, pz, pe, pr):
        p.free()
print(f"{cases} cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
