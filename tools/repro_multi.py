import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from diaglib_amd import capi
ctx = capi.Context()
rng = np.random.default_rng(1)
n = 4000
ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
mv = capi.fn_address("dla_synth_matvec")
for knob in (12, 13):
    ctx.set_option(100 + 6, knob)
    for k in (1, 3, 13):
        nb = 6
        ld = nb * k
        x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, k)))[0])
        basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - k))])))
        abasis = ctx.panel(np.zeros((n, ld), order="F"))
        ctx.synth_matvec(basis.col(0, k), abasis.col(0, k))
        hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
        b = basis.download(); ab = abasis.download()
        hraw[:k, :k] = b[:, :k].T @ ab[:, :k]; h[:k, :k] = hraw[:k, :k]
        for blk in range(1, nb):
            m = blk * k
            # new block: partly inside span(X) so that the projection matters
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            bb = basis.download(); bb[:, m:m + k] = u
            basis = ctx.panel(np.asfortranarray(bb))
            h4 = ctx.expand_project(4, basis, abasis, m, k, mv, 0.0)
            p = ctx.pending_block(m, k); applied = ctx.pending_applied
            p0 = p.copy()
            h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=applied)
            b = basis.download(); ab = abasis.download()
            l = m + k
            v = b[:, :l] @ dmat[:l, :l]
            href = v.T @ (ab[:, :l] @ dmat[:l, :l])
            print(knob, k, blk, "E'", np.abs(p0[:m]).max(), "T-I", np.abs(p0[m:] - np.eye(k)).max(), "orth(VD)", np.abs(v.T @ v - np.eye(l)).max(),
                  "stored orth", np.abs(b[:, :l].T @ b[:, :l] - np.eye(l)).max(), "h err", np.abs(np.triu(h[:l, :l] - href)).max() / np.abs(href).max(), flush=True)
