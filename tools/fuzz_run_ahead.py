#!/usr/bin/env python3
"""Run-ahead of the expansion step (dla_expand_project / dla_expand_project_metric, DESIGN 4, docs/HISTORY.md 11.3) on and off: the same Davidson / LOBPCG
solves (standard and generalised) on the device-resident benchmark operators must give the SAME BITS either way -- eigenvalues, eigenvectors, iteration and operator
column counts -- because the run-ahead changes when the host learns the orthogonalisation's outcome, not one kernel's input.
    python tools/fuzz_run_ahead.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402


def solve(ctx, solver, n, t, m, tol, guess_rows, ahead, shift):
    ctx.set_option(capi.OPT_RUN_AHEAD, 1 if ahead else 0)
    ev = ctx.panel(n, m)
    if guess_rows == 0:
        g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        ev.upload(g)
    else:
        ctx.fill_guess(ev, 2, support_rows=guess_rows)
    mv, pc, bv = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd"), capi.fn_address("dla_synth_metric")
    s0 = ctx.stats()["host_syncs"]
    if solver == "davidson":
        eig, _, ok, info = ctx.davidson_driver(n, t, m, 300, tol, 8, 0.0, mv, pc, ev)
    elif solver == "gen_david":
        eig, _, ok, info = ctx.gen_david_driver(n, t, m, 300, tol, 8, 0.0, mv, pc, bv, ev)
    elif solver == "gen_lobpcg":
        eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, ev, bvec=bv)
    else:
        eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 300, tol, shift, mv, pc, ev)
    syncs = ctx.stats()["host_syncs"] - s0
    x = ev.download()
    ev.free()
    return np.array(eig[:t]), x[:, :t], ok, info, syncs


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    ctx = capi.Context()
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.set_option(capi.OPT_EVEC_ON_DEVICE, 1)
    bad = 0
    for it in range(cases):
        n = int(rng.integers(3000, 60000)); n += int(rng.integers(0, 2))
        t = int(rng.integers(1, 30)); m = min(48, t + int(rng.integers(1, 8)))
        solver = str(rng.choice(["davidson", "lobpcg", "gen_david", "gen_lobpcg"]))
        guess_rows = int(rng.choice([0, 0, 400]))
        shift = float(rng.choice([0.0, 0.0, 0.5])) if solver.endswith("lobpcg") else 0.0
        ctx.set_shard(n, 0)
        ctx.synth_setup(n, 0, n)
        res = [solve(ctx, solver, n, t, m, 1e-9, guess_rows, ahead, shift) for ahead in (True, False, True)]
        (e1, x1, ok1, i1, s1), (e0, x0, ok0, i0, s0), (e2, x2, ok2, i2, s2) = res
        same = (np.array_equal(e1, e0) and np.array_equal(x1, x0) and i1 == i0 and ok1 == ok0 and
                np.array_equal(e2, e0) and np.array_equal(x2, x0) and i2 == i0)
        if not same and solver.startswith("gen_"):
            # with a metric the run-ahead also moves b_ortho's k x k factorisation from the host to the device (same operation
            # order, but the host's compiler may contract multiply-adds differently): the two runs of the SAME mode must agree bit
            # for bit, the two modes to rounding with the same history
            same = (np.array_equal(e1, e2) and np.array_equal(x1, x2) and i1 == i2 and i1["iters"] == i0["iters"] and ok1 == ok0 and
                    np.abs(e1 - e0).max() <= 1e-10 * np.abs(e0).max())
        print(f"case {it:3d} {solver:8s} n={n:6d} roots={t:2d} n_max={m:2d} guess_rows={guess_rows:4d} shift={shift}: ok={ok1} iters={i1['iters']:3d} "
              f"restarts={i1['restarts']} waits {s1}/{s0}/{s2} (ahead/off/ahead) {'same bits' if same else 'DIFFERENT'}", flush=True)
        bad += 0 if same else 1
    ctx.set_shard(-1, 0)
    print(f"{cases} cases, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
