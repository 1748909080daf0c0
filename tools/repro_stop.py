import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from diaglib_amd import capi
EPS = np.finfo(float).eps
ctx = capi.Context(); rng = np.random.default_rng(5)
n, k, nb = 5000, 8, 7
mv = capi.fn_address("dla_synth_matvec")
for maxit, kind in ((1, "dup"), (2, "dup"), (2, "near"), (3, "dup")):
    ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ld = nb * k
    x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, k)))[0])
    basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - k))]))); abasis = ctx.panel(np.zeros((n, ld), order="F"))
    ctx.synth_matvec(basis.col(0, k), abasis.col(0, k))
    hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
    b = basis.download(); ab = abasis.download()
    hraw[:k, :k] = b[:, :k].T @ ab[:, :k]; h[:k, :k] = hraw[:k, :k]
    ctx.basis_sync(0, 0); ctx.basis_sync(0, k, dmat)
    print("== maxit", maxit, kind, flush=True)
    try:
        for blk in range(1, nb):
            m = blk * k
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            if blk >= 4:
                if kind == "dup": u[:, -1] = u[:, 0] * (1.0 + 1e-13) + 1e-14 * rng.standard_normal(n)
                else: u[:, 1:] = u[:, :1] + 1e-9 * rng.standard_normal((n, k - 1))
                ctx.set_option(capi.OPT_ORTHO_MAXIT, maxit)
            basis.col(m, k).upload(np.asfortranarray(u))
            s0 = ctx.stats()["host_syncs"]
            h4 = ctx.expand_project(5, basis, abasis, m, k, mv, 0.0)
            print("   block", blk, "host waits", ctx.stats()["host_syncs"] - s0, flush=True)
            ctx.set_option(capi.OPT_ORTHO_MAXIT, 10)
            p = ctx.pending_block(m, k); h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied); ctx.basis_sync(m, k, dmat)
            b = basis.download()
        v = b @ dmat
        print("   ortho", np.abs(v.T @ v - np.eye(ld)).max(), "D nontrivial", not np.array_equal(dmat, np.eye(ld)), flush=True)
    except Exception as e:
        print("   exception", str(e)[:200], flush=True)
    ctx.set_option(capi.OPT_ORTHO_MAXIT, 10)
