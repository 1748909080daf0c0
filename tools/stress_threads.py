#!/usr/bin/env python3
"""Several host threads, each with its own context, run different drivers at the same time on one GPU (standard, generalised,
linear-response, the sparse sample operator); every result must equal, to the bit, what the same call returns when it runs alone.

    python tools/stress_threads.py [threads] [solves per thread] [seed]"""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import scipy.sparse as sp  # noqa: E402
from diaglib_amd import capi  # noqa: E402

nthreads = int(sys.argv[1]) if len(sys.argv) > 1 else 6
per = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
KINDS = ["davidson", "lobpcg", "gen_david", "lobpcg_gen", "caslr_eff", "caslr", "spmm"]


def solve(c, spec):
    kind, n, t, m = spec["kind"], spec["n"], spec["t"], spec["m"]
    c.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    A = capi.fn_address
    if kind == "spmm":
        i = np.arange(n, dtype=np.float64)
        diags, offs = [2.0 + i / 50.0], [0]
        for d in range(1, 4):
            v = 0.3 / d * np.cos(i[:n - d] + d); diags += [v, v]; offs += [d, -d]
        c.spmm_setup(sp.diags(diags, offs, shape=(n, n), format="csr"))
        g = np.asfortranarray(np.random.default_rng(spec["gseed"]).random((n, m)) - 0.5)
        return c.davidson_driver(n, t, m, 200, 1e-9, 20, 0.0, A("dla_spmm_matvec"), A("dla_spmm_precnd"), c.panel(g))[0]
    c.synth_setup(n, 0, n, 4, spec["sigma"])
    if kind in ("caslr_eff", "caslr"):
        g = np.zeros((2 * n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
        fns = [A(k) for k in ("dla_synth_apbmul", "dla_synth_ambmul", "dla_synth_spdmul", "dla_synth_smdmul",
                              "dla_synth_lrprec1" if kind == "caslr" else "dla_synth_lrprec2")]
        return (c.caslr_driver if kind == "caslr" else c.caslr_eff_driver)(n, t, m, 200, 1e-9, 10, *fns, c.panel(g))[0]
    g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
    mv, pc, bv = A("dla_synth_matvec"), A("dla_synth_precnd"), A("dla_synth_metric")
    if kind == "davidson":
        return c.davidson_driver(n, t, m, 200, 1e-10, 20, 0.0, mv, pc, c.panel(g))[0]
    if kind == "lobpcg":
        return c.lobpcg_driver(n, t, m, 200, 1e-10, 0.0, mv, pc, c.panel(g))[0]
    if kind == "gen_david":
        return c.gen_david_driver(n, t, m, 200, 1e-10, 20, 0.0, mv, pc, bv, c.panel(g))[0]
    return c.lobpcg_driver(n, t, m, 200, 1e-10, 0.0, mv, pc, c.panel(g), bvec=bv)[0]


rng = np.random.default_rng(seed)
specs = [[dict(kind=str(rng.choice(KINDS)), n=int(rng.integers(30_000, 150_000)) * 2, t=int(rng.integers(2, 9)), m=0,
               sigma=float(rng.choice([0.25, 0.5])), gseed=int(rng.integers(1, 1000))) for _ in range(per)] for _ in range(nthreads)]
for th in specs:
    for s in th:
        s["m"] = s["t"] + int(rng.integers(1, 6))
# alone
c0 = capi.Context()
alone = [[solve(c0, s).copy() for s in th] for th in specs]
c0.trim()
got = [[None] * per for _ in range(nthreads)]
errs = []


def work(i):
    try:
        c = capi.Context()
        for j, s in enumerate(specs[i]):
            got[i][j] = solve(c, s).copy()
            c.trim()
    except Exception as e:   # noqa: BLE001
        errs.append((i, repr(e)))


ths = [threading.Thread(target=work, args=(i,)) for i in range(nthreads)]
[t.start() for t in ths]
[t.join() for t in ths]
bad = len(errs)
for e in errs:
    print("EXCEPTION in thread", e, flush=True)
for i in range(nthreads):
    for j in range(per):
        if got[i][j] is None or not np.array_equal(got[i][j], alone[i][j]):
            bad += 1
            print("MISMATCH", specs[i][j], None if got[i][j] is None else float(np.abs(got[i][j] - alone[i][j]).max()), flush=True)
print(f"{nthreads} threads x {per} solves: {bad} problems", flush=True)
sys.exit(1 if bad else 0)
