#!/usr/bin/env python3
"""Generalised drivers at full size with a device-resident metric callback (B = I as a device copy): timing and
agreement with the standard drivers.   python tools/gen_probe.py [n]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
t, m = 8, 13
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")


@capi.MATVEC_T
def bvec(pn, pm, px, pax):
    nn, mm = pn[0], pm[0]
    ctx.lib.dla_copy(ctx.h, C.cast(pax, C.c_void_p), C.cast(px, C.c_void_p), 8 * nn * mm)


bv = C.cast(bvec, C.c_void_p).value
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
ev = ctx.panel(g)
res = {}
for name in ("davidson", "gen_david", "lobpcg", "lobpcg_gen"):
    for rep in range(2):
        ev.upload(g)
        ctx.sync()
        t0 = time.perf_counter()
        if name == "davidson":
            eig, _, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, ev)
        elif name == "gen_david":
            eig, _, ok, info = ctx.gen_david_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, bv, ev)
        elif name == "lobpcg":
            eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 200, 2e-13, 0.0, mv, pc, ev)
        else:
            eig, _, ok, info = ctx.lobpcg_driver(n, t, m, 200, 2e-13, 0.0, mv, pc, ev, bvec=bv)
        dt = time.perf_counter() - t0
    res[name] = eig[:t].copy()
    print(f"{name:11s} {dt * 1e3:8.2f} ms  iters {info['iters']:3d} restarts {info['restarts']} ok {ok}", flush=True)
print("gen_david vs davidson", np.abs(res["gen_david"] - res["davidson"]).max())
print("lobpcg_gen vs lobpcg ", np.abs(res["lobpcg_gen"] - res["lobpcg"]).max())
