import sys, numpy as np
sys.path.insert(0, '.')
from diaglib_amd import capi
from oracle.pyoracle import Oracle
ctx = capi.Context(); oracle = Oracle()
cases = [("davidson", 1001, 4, 8), ("lobpcg", 1001, 4, 8), ("davidson", 3000, 16, 21), ("lobpcg", 3000, 16, 21), ("davidson", 2500, 32, 37), ("lobpcg", 2500, 32, 37)]
import re
src = open('tests/test_solver_gpu.py').read()
m = re.search(r'parametrize\("solver,n,n_targ,n_max", \[(.*?)\]\)\ndef test_block_widths', src, re.S)
cases = eval('[' + m.group(1) + ']')
for solver, n, n_targ, n_max in cases:
    oracle.dense_setup(n)
    mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
    g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
    out = []
    for knob in (0, 7):
        ctx.set_option(100 + 6, knob)
        if solver == "davidson":
            eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
        else:
            eig, v, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
        out.append((info["iters"], info["matvec_cols"]))
    ctx.set_option(100 + 6, 0)
    if solver == "davidson":
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 100, 1e-8, 20, 0.0, mv, pc, g)
    else:
        eo, vo, oko, tr = oracle.lobpcg(n, n_targ, n_max, 100, 1e-8, 0.0, mv, pc, g)
    print(solver, n, n_targ, n_max, "hip default", out[0], "lead_once off", out[1], "oracle", (tr.iters, tr.matvec_cols), flush=True)
