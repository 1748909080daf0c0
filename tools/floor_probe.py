#!/usr/bin/env python3
"""Where the residual of a converged LOBPCG / Davidson run stops falling, and what the locking rule (reference
diaglib.f90:446-455, 1737-1746: rms < tol and max < 10 tol) does there -- for the HIP path, the oracle's C restatement and
the UNMODIFIED reference (oracle/_ref), on the benchmark operator A = diag(i+1) + 0.5 W W^T.

    python tools/floor_probe.py --n 1000000 --roots 32 --solver lobpcg --iters 40 [--impl hip,oracle,reference]

Every implementation runs ONCE with a tolerance it cannot meet (1e-30) for --iters iterations, verbose, in a child process; the
per-iteration table (the reference's own trace format) is parsed and the history of max_i rms_i / max_i max|r_i| over the
wanted roots is printed.  The floor is the level the history settles at; a tolerance is safe when 10 tol sits well above the
floor of max|r| (and tol above the floor of rms).  TEST / MEASUREMENT AID: imports oracle/ as the checker."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import numpy as np
sp = json.loads({spec!r})
n, t, m = sp["n"], sp["roots"], sp["n_max"]
impl = sp["impl"]
if impl == "hip":
    from diaglib_amd import capi
    ctx = capi.Context()
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.synth_setup(n, 0, n)
    mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    ev = ctx.panel(g)
    if sp["solver"] == "lobpcg":
        ctx.lobpcg_driver(n, t, m, sp["iters"], sp["tol"], 0.0, mv, pc, ev, verbose=True)
    else:
        ctx.davidson_driver(n, t, m, sp["iters"], sp["tol"], 20, 0.0, mv, pc, ev, verbose=True)
else:
    from oracle.pyoracle import Oracle, Reference
    o = Oracle()
    o.synth_setup(n, 0, n)
    mv, pc = o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd")
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    r = Reference() if impl == "reference" else o
    if sp["solver"] == "lobpcg":
        r.lobpcg(n, t, m, sp["iters"], sp["tol"], 0.0, mv, pc, g, verbose=True)
    else:
        r.davidson(n, t, m, sp["iters"], sp["tol"], 20, 0.0, mv, pc, g, verbose=True)
    if impl == "reference" and hasattr(r.lib, "ref_flush"):
        r.lib.ref_flush()
sys.stdout.flush()
"""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--roots", type=int, default=32)
    ap.add_argument("--solver", default="lobpcg", choices=["lobpcg", "davidson"])
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--tol", type=float, default=1e-30)
    ap.add_argument("--impl", default="hip,oracle,reference")
    ap.add_argument("--out", default=None, help="write the histories as JSON")
    args = ap.parse_args()
    from make_golden import parse_trace
    n_max = min(2 * args.roots, args.roots + 5)
    res = {}
    for impl in args.impl.split(","):
        spec = dict(n=args.n, roots=args.roots, n_max=n_max, solver=args.solver, iters=args.iters, tol=args.tol, impl=impl)
        p = subprocess.run([sys.executable, "-c", WORKER.format(root=ROOT, spec=json.dumps(spec))], capture_output=True, text=True)
        if p.returncode != 0:
            print(impl, "failed:", p.stderr[-1500:])
            continue
        tr = parse_trace(p.stdout, args.roots)
        rms, rmx = tr["rms"].max(1), tr["rmax"].max(1)
        res[impl] = dict(iters=int(tr["iters"]), rms=[float(x) for x in rms], rmax=[float(x) for x in rmx],
                         locked=[int(x) for x in tr["done"].sum(1)])
        print(f"{impl:9s} n={args.n} roots={args.roots} {args.solver}: {tr['iters']} iterations")
        for i in range(tr["iters"]):
            print(f"   it {i + 1:3d}  max rms {rms[i]:9.2e}  max |r| {rmx[i]:9.2e}  locked {int(tr['done'][i].sum()):3d}")
        tail = slice(max(0, tr["iters"] - 10), tr["iters"])
        print(f"   floor (median of the last 10): rms {np.median(rms[tail]):.2e}  max|r| {np.median(rmx[tail]):.2e}")
    if args.out:
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
