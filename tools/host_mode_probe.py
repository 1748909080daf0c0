#!/usr/bin/env python3
"""Drop-in mode (HOST callbacks, host evec) of the benchmark workload: what an unmodified Fortran caller gets.
The operator is the oracle's C implementation of the benchmark operator running on the host cores; every block
crosses PCIe twice per callback.  Prints the PCIe-inclusive solve time next to the device-resident one.
    python tools/host_mode_probe.py [n]"""
import os
import sys
import time

os.environ.setdefault("OMP_NUM_THREADS", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from diaglib_amd import capi  # noqa: E402
from oracle.pyoracle import Oracle  # noqa: E402  (tools/ is measurement infrastructure, like bench.py's cpu_baseline)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
t, m = 8, 13
ctx = capi.Context()
o = Oracle()
o.synth_setup(n, 0, n)
ctx.synth_setup(n, 0, n)
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0

ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
mv, pc = o.fn("orc_synth_matvec"), o.fn("orc_synth_precnd")
for chunks in (0, 1, 2, 3, 4):
    ctx.set_option(capi.OPT_STAGE_CHUNKS, chunks)
    for rep in range(2):
        ctx.reset_stats()
        t0 = time.perf_counter()
        eig, v, ok, info = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, mv, pc, g)
        dt_host = time.perf_counter() - t0
    print(f"host callbacks + host evec, stage chunks {chunks} (0 = automatic): {dt_host * 1e3:8.1f} ms per solve, "
          f"{info['iters']} iterations, ok={ok}")
ctx.set_option(capi.OPT_STAGE_CHUNKS, 0)

ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ev = ctx.panel(g)
dmv, dpc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
for rep in range(2):
    ev.upload(g)
    t0 = time.perf_counter()
    eig2, _, ok2, info2 = ctx.davidson_driver(n, t, m, 200, 2e-13, 20, 0.0, dmv, dpc, ev)
    dt_dev = time.perf_counter() - t0
print(f"device callbacks + device evec: {dt_dev * 1e3:8.1f} ms per solve, {info2['iters']} iterations, ok={ok2}")
print("eigenvalue difference", np.abs(eig[:t] - eig2[:t]).max())
blk = 8.0 * n * m
print(f"PCIe volume per solve (host mode): {(2 * info['matvec_cols'] / m + 2 * info['iters']) * blk / 1e9:.2f} GB")
if os.environ.get("DIAGLIB_AMD_HOSTTIME"):
    ev.free()
    ctx.lib.dla_destroy(ctx.h)       # prints the per-entry-point wall times
