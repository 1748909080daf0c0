import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from diaglib_amd import capi
from oracle import pyoracle
oracle = pyoracle.Oracle() if hasattr(pyoracle, "Oracle") else pyoracle.load()
ctx = capi.Context()
n = 800
oracle.dense_setup(n)
mv, pc = oracle.fn("orc_dense_matvec"), oracle.fn("orc_dense_precnd")
for knob in (12, 13):
    for n_targ, n_max in ((1, 1), (3, 3), (1, 6)):
        g = np.zeros((n, n_max), order="F"); g[np.arange(n_max), np.arange(n_max)] = 1.0
        ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
        ctx.set_option(100 + 6, knob)
        eig, v, ok, info = ctx.davidson_driver(n, n_targ, n_max, 200, 1e-9, 20, 0.0, mv, pc, g)
        eo, vo, oko, tr = oracle.davidson(n, n_targ, n_max, 200, 1e-9, 20, 0.0, mv, pc, g)
        print(knob, n_targ, n_max, eig[:n_targ], eo[:n_targ], ok, info, tr.iters, flush=True)
