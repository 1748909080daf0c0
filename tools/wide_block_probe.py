#!/usr/bin/env python3
"""Drivers with more than 48 columns per block (50 wanted roots, n_max = 55) against the oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from diaglib_amd import capi
from oracle.pyoracle import Oracle
o = Oracle(); ctx = capi.Context()
n, t, m = 3000, 50, 55
o.dense_setup(n); mv, pc = o.fn("orc_dense_matvec"), o.fn("orc_dense_precnd")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
idx = np.arange(1, n + 1.0); a = 1 / (idx[:, None] + idx[None, :]); np.fill_diagonal(a, idx + 1); w = np.linalg.eigvalsh(a)[:t]
bad = 0
for solver in ("davidson", "lobpcg"):
    if solver == "davidson":
        e, v, ok, info = ctx.davidson_driver(n, t, m, 100, 1e-9, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = o.davidson(n, t, m, 100, 1e-9, 20, 0.0, mv, pc, g)
    else:
        e, v, ok, info = ctx.lobpcg_driver(n, t, m, 100, 1e-9, 0.0, mv, pc, g)
        eo, vo, oko, tr = o.lobpcg(n, t, m, 100, 1e-9, 0.0, mv, pc, g)
    err = np.abs(e[:t] - w).max()
    print(solver, "ok", ok, oko, "iters", info["iters"], tr.iters, "max eig err", err, flush=True)
    bad += (not ok) or err > 1e-8
sys.exit(1 if bad else 0)
