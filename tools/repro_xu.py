import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from diaglib_amd import capi
ctx = capi.Context()
rng = np.random.default_rng(1)
n = 4000
ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
for knob in (12, 13):
    ctx.set_option(100 + 6, knob)
    for m, k in ((2, 1), (3, 3), (26, 13), (5, 1)):
        x = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, m)))[0])
        u = np.asfortranarray(rng.standard_normal((n, k)))
        px = ctx.panel(x); pax = ctx.panel(np.zeros_like(x)); ctx.synth_matvec(px, pax); ax = pax.download()
        for rep in range(2):
            basis = ctx.panel(np.asfortranarray(np.hstack([x, u]))); abasis = ctx.panel(np.asfortranarray(np.hstack([ax, np.zeros((n, k))])))
            h4 = ctx.expand_project(4, basis, abasis, m, k, capi.fn_address("dla_synth_matvec"), 0.0)
            p = ctx.pending_block(m, k); applied = ctx.pending_applied
            b4 = basis.download()
            l = m + k
            hraw = np.zeros((l, l), order="F"); dmat = np.asfortranarray(np.eye(l)); h = np.zeros((l, l), order="F")
            hraw[:m, :m] = x.T @ ax; h[:, m:] = h4
            p0 = p.copy()
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=applied)
            w = b4 @ p
            print(knob, m, k, rep, "applied", applied, "E'max", np.abs(p0[:m]).max(), "T", np.abs(p0[m:] - np.eye(k)).max(), "orth", np.abs(w.T @ w - np.eye(k)).max(), "xTw", np.abs(x.T @ w).max(), flush=True)
