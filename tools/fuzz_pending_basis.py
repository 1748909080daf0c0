#!/usr/bin/env python3
"""Randomised bases grown block by block through dla_expand_project modes 4 / 5 (the Davidson drivers' expansion step with the
closing pass left to the caller's small matrices; docs/HISTORY.md 14.1 / 14.1b): random row counts (odd ones take the sweep-per-update
schedule), block widths 1 .. 16, up to 320 columns, random / nearly dependent / inside-span(X) / tiny-norm blocks, every schedule
knob.  Checked: (panel D)^T (panel D) = I to 100 eps and h = (panel D)^T A (panel D) to 1e-12.

    python tools/fuzz_pending_basis.py [cases] [seed] [only this case]      (FUZZ_WIDE=1 or a fourth argument `wide`: blocks of 17 .. 40 columns, mode 4;
                                                                             a fourth argument `mixed`: mode 5 with a width of its own per block,
                                                                             1 .. 40 columns -- wide blocks are finished in memory, by the device chain
                                                                             while nothing is pending in front of them and by the host-driven loop with
                                                                             D D^T otherwise (round-5 advisor), bases of up to 400 columns: beyond the
                                                                             320 of the device copy of D the host-driven loop takes every block)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
EPS = np.finfo(np.float64).eps
ctx = capi.Context()
mv = capi.fn_address("dla_synth_matvec")
bad = 0
only = int(sys.argv[3]) if len(sys.argv) > 3 else -1          # run this case alone
for it in range(cases):
    if only >= 0 and it != only:
        continue
    rng = np.random.default_rng([seed, it])
    k = int(rng.integers(1, 17))
    nb = int(rng.integers(3, max(4, min(26, 320 // k) + 1)))
    mode = int(rng.choice([4, 5, 5]))
    if os.environ.get("FUZZ_WIDE") or (len(sys.argv) > 4 and sys.argv[4] == "wide"):             # blocks of two and three column tiles (the LDS-loop k x k step): mode 4 only
        k = int(rng.integers(17, 41)); nb = int(rng.integers(3, 11)); mode = 4
    mixed = len(sys.argv) > 4 and sys.argv[4] == "mixed"
    widths = [k] * nb
    if mixed:
        mode = 5
        nb = int(rng.integers(3, 14))
        widths = [int(rng.choice([rng.integers(1, 17), rng.integers(1, 17), rng.integers(17, 41)])) for _ in range(nb)]
        while sum(widths) > 400:
            widths.pop()
        nb = len(widths); k = max(widths)
    n = int(rng.integers(max(600, 3 * sum(widths)), 9000))
    knob = int(rng.choice([0, 0, 12, 13, 15, 16]))
    spec = dict(n=n, k=k, nb=nb, mode=mode, knob=knob)
    if mixed:
        spec["widths"] = widths
    try:
        ctx.set_shard(n, 0); ctx.synth_setup(n, 0, n); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
        ctx.set_option(100 + 6, knob)
        ld = sum(widths)
        k0 = widths[0]
        x0 = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, k0)))[0])
        basis = ctx.panel(np.asfortranarray(np.hstack([x0, np.zeros((n, ld - k0))])))
        abasis = ctx.panel(np.zeros((n, ld), order="F"))
        ctx.synth_matvec(basis.col(0, k0), abasis.col(0, k0))
        hraw = np.zeros((ld, ld), order="F"); dmat = np.asfortranarray(np.eye(ld)); h = np.zeros((ld, ld), order="F")
        b = basis.download(); ab = abasis.download()
        hraw[:k0, :k0] = b[:, :k0].T @ ab[:, :k0]; h[:k0, :k0] = hraw[:k0, :k0]
        if mode == 5:
            ctx.basis_sync(0, 0); ctx.basis_sync(0, k0, dmat)
        kinds = []
        for blk in range(1, nb):
            m = sum(widths[:blk]); k = widths[blk]
            kind = str(rng.choice(["random", "random", "inside", "dependent", "tiny", "scaled"]))
            kinds.append(kind)
            u = b[:, :m] @ rng.standard_normal((m, k)) + 0.3 * rng.standard_normal((n, k))
            if kind == "inside":
                u = b[:, :m] @ rng.standard_normal((m, k)) + 10.0 ** rng.integers(-9, -3) * rng.standard_normal((n, k))
            elif kind == "dependent" and k > 1:
                u[:, -1] = u[:, 0] * (1.0 + 1e-13) + 1e-14 * rng.standard_normal(n)
            elif kind == "tiny":
                u *= 10.0 ** rng.integers(-12, -6)
            elif kind == "scaled":
                u *= 10.0 ** rng.integers(-6, 7, size=k)[None, :]
            basis.col(m, k).upload(np.asfortranarray(u))
            h4 = ctx.expand_project(mode, basis, abasis, m, k, mv, 0.0)
            p = ctx.pending_block(m, k)
            h[:m + k, m:m + k] = h4
            ctx.basis_admit(m, k, p, hraw, dmat, h, applied=ctx.pending_applied)
            if mode == 5:
                ctx.basis_sync(m, k, dmat)
            b = basis.download()
        ab = abasis.download()
        v = b @ dmat
        e1 = float(np.abs(v.T @ v - np.eye(ld)).max())
        href = v.T @ (ab @ dmat)
        e2 = float(np.abs(np.triu(h - href)).max() / np.abs(href).max())
        good = e1 < 100 * EPS and e2 < 1e-12
        print(("ok  " if good else "FAIL"), dict(spec, case=it), dict(ortho=e1, h=e2), "" if good else kinds, flush=True)
        bad += 0 if good else 1
    except Exception as ex:   # noqa: BLE001
        bad += 1
        print("FAIL (exception)", spec, str(ex)[:300], flush=True)
    finally:
        ctx.set_option(100 + 6, 0); ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0); ctx.set_shard(-1, 0)
print(f"{cases} pending-basis cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
