#!/usr/bin/env python3
"""Round-trip cost of one host-visible reduction on a tiny input (two small kernels + the host wait)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from diaglib_amd import capi
ctx = capi.Context()
x = ctx.panel(np.ones((64, 1)))
for v in (0, 2):
    ctx.set_option(106, v)
    for _ in range(200): ctx.nrm2(x)
    t0 = time.perf_counter()
    N = 5000
    for _ in range(N): ctx.nrm2(x)
    dt = (time.perf_counter() - t0) / N
    print(f"knob6={v}: {dt * 1e6:.2f} us per tiny reduction (python call overhead included)")
g = ctx.panel(np.ones((64, 13)))
t0 = time.perf_counter()
for _ in range(N): ctx.gram(g, g)
print(f"tiny gram: {(time.perf_counter() - t0) / N * 1e6:.2f} us")
