#!/usr/bin/env python3
"""ortho_cd / ortho_vs_x on blocks that have NO full rank: duplicate columns, zero columns, columns inside span(X), low-rank
products, blocks whose support is a handful of rows (rounding noise has nowhere to go), extreme column scales.
Accepted outcomes: success with U orthonormal, orthogonal to X and containing the independent part of the input -- what the
reference's `ortho` fallback (dgeqrf + dorgqr) delivers -- or an error return.  Never: success with anything else, NaNs, a hang.

    python tools/fuzz_degenerate.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
EPS = np.finfo(np.float64).eps
ctx = capi.Context()
bad = 0
errors = 0
kinds = ["duplicate", "zero", "in_span_x", "low_rank", "few_rows", "scales", "mixed"]
for it in range(cases):
    ctx.set_option(100 + 6, (0, 12, 13)[it % 3])      # sweep schedule of ortho_vs_x: shipped choice / five-sweep / three-pass always
    kind = kinds[it % len(kinds)]
    k = int(rng.choice([rng.integers(2, 17), rng.integers(17, 40)]))
    m = int(rng.choice([0, rng.integers(1, 30), rng.integers(30, 120)]))
    if kind == "in_span_x" and m == 0:
        m = int(rng.integers(2, 30))
    n = int(rng.integers(max(m + k, 8) * 3, max(m + k, 8) * 3 + 5000))
    n += n % 2
    if kind == "few_rows":
        # X = unit vectors, U lives in the rows of X and a few more
        rows = m + int(rng.integers(1, k))
        x = np.zeros((n, m)); x[np.arange(m), np.arange(m)] = 1.0
        u = np.zeros((n, k)); u[:rows, :] = rng.standard_normal((rows, k))
    else:
        x = np.linalg.qr(rng.standard_normal((n, max(m, 1))))[0][:, :m]
        u = rng.standard_normal((n, k))
        if m:
            u += x @ (rng.standard_normal((m, k)) * rng.choice([0.0, 1.0, 30.0]))
    if kind in ("duplicate", "mixed"):
        j = int(rng.integers(1, k)); u[:, j] = u[:, int(rng.integers(0, j))]
    if kind in ("zero", "mixed"):
        u[:, int(rng.integers(0, k))] = 0.0
    if kind == "in_span_x":
        j = int(rng.integers(0, k)); u[:, j] = x @ rng.standard_normal(m)
    if kind == "low_rank":
        r = int(rng.integers(1, k)); u = u[:, :r] @ rng.standard_normal((r, k))
    if kind == "scales":
        u *= (10.0 ** rng.choice([-150.0, -100.0, -20.0, 0.0, 20.0, 100.0, 140.0], size=k))[None, :]
    x = np.asfortranarray(x); u = np.asfortranarray(u)
    # the part of the input outside span(X), as an orthonormal basis of its numerical range
    p = u - x @ (x.T @ u) if m else u.copy()
    nrm = np.linalg.norm(u, axis=0); nrm[nrm == 0.0] = 1.0          # (relative to the INPUT column: a column inside span(X) leaves noise)
    uu, sv, _ = np.linalg.svd(p / nrm, full_matrices=False)
    indep = uu[:, sv > 1e-6]
    contiguous = bool(m) and rng.random() < 0.6
    try:
        if m and contiguous:
            panel = ctx.panel(np.asfortranarray(np.hstack([x, u])))
            ctx.ortho_vs_x(panel.col(0, m), panel.col(m, k))
            got = panel.col(m, k).download(); panel.free()
        elif m:
            px, pu = ctx.panel(x), ctx.panel(u)
            ctx.ortho_vs_x(px, pu)
            got = pu.download(); px.free(); pu.free()
        else:
            pu = ctx.panel(u)
            _, ok = ctx.ortho_cd(pu)
            got = pu.download(); pu.free()
            if not ok:                          # ortho_cd reports its failure through `ok` (reference diaglib.f90:3185: the caller runs `ortho`)
                pq = ctx.panel(got)
                ctx.ortho_qr(pq)
                got = pq.download(); pq.free()
    except Exception as e:                      # an error return is an accepted outcome
        errors += 1
        continue
    res = {"orthonormal": np.abs(got.T @ got - np.eye(k)).max() / (50 * EPS)}
    if m:
        res["x_orth"] = np.abs(x.T @ got).max() / (50 * EPS)
    res["contains_input"] = np.abs(indep - got @ (got.T @ indep)).max() / 1e-8 if indep.shape[1] else 0.0
    worst = max(res.values())
    if worst > 1.0 or not np.isfinite(worst):
        bad += 1
        print("FAIL", dict(kind=kind, n=n, m=m, k=k, contiguous=contiguous), {a: float(b) for a, b in res.items()}, flush=True)
print(f"{cases} cases, {bad} failures, {errors} error returns", flush=True)
sys.exit(1 if bad else 0)
