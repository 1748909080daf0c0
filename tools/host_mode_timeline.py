#!/usr/bin/env python3
"""One drop-in (host-callback, host-evec) solve of the benchmark workload for a timeline: under rocprofv3
(--hip-trace --memory-copy-trace --kernel-trace) and / or with DIAGLIB_AMD_HOSTTIME=1 (per-entry-point wall times).
    python3 tools/host_mode_timeline.py [n] [chunks]"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402  (measurement infrastructure: the host operator is the oracle's)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t, m = 8, 13
ctx = capi.Context()
o = Oracle()
o.synth_setup(n, 0, n)
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
ctx.set_option(capi.OPT_STAGE_CHUNKS, chunks)
mv, pc = o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd")
for rep in range(2):
    t0 = time.perf_counter()
    eig, v, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, g)
    dt = time.perf_counter() - t0
    print(f"solve {rep}: {dt * 1e3:.1f} ms, {info['iters']} iterations, ok={ok}", flush=True)
# the caller's routines alone, on host arrays of the same shape (what the 17 callbacks of a solve cost without any transfer)
x = np.asfortranarray(np.random.default_rng(0).standard_normal((n, m)))
y = np.zeros_like(x)
import ctypes as C
MV = C.CFUNCTYPE(None, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_void_p)
f = C.cast(mv, MV)
nn, mm = C.c_int(n), C.c_int(m)
for rep in range(3):
    t0 = time.perf_counter()
    f(C.byref(nn), C.byref(mm), x.ctypes.data, y.ctypes.data)
    dtm = time.perf_counter() - t0
print(f"caller's matvec alone on a {n} x {m} host block: {dtm * 1e3:.2f} ms")
if os.environ.get("DIAGLIB_AMD_HOSTTIME"):
    ctx.lib.dla_destroy(ctx.h)
