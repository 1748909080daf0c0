import sys, os, json
sys.path.insert(0, '.')
import numpy as np
from diaglib_amd import capi
n, n_targ = 10_000_000, 32
n_max = 37
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
ctx.synth_setup(n, 0, n)
mv, pc = capi.fn_address("dla_synth_matvec"), capi.fn_address("dla_synth_precnd")
guess = np.zeros((n, n_max), order="F")
for j in range(n_max): guess[j, j] = 1.0
g_dev = ctx.panel(guess); ev = ctx.panel(n, n_max)
for knob in (0, 9):
    ctx.set_option(100 + 6, knob)
    for rep in range(2):
        ctx.lib.dla_copy(ctx.h, ev.ptr, g_dev.ptr, 8 * n * n_max)
        ctx.reset_stats(); ctx.set_option(capi.OPT_PROFILE, 1 if rep == 1 else 0)
        eig, _, ok, info = ctx.lobpcg_driver(n, n_targ, n_max, 400, 1e-12, 0.0, mv, pc, ev)
    ks = ctx.kernel_stats(); ctx.set_option(capi.OPT_PROFILE, 0)
    print("knob", knob, "iters", info["iters"])
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"]):
        if v["ms"] > 1.0 and ("gram" in k or "gemm" in k):
            print(f"  {v['ms']:8.2f} ms {v['launches']:4d} x {v['ms']/max(1,v['launches'])*1e3:8.1f} us  {v['alg_bytes']/max(v['ms'],1e-9)/1e6:7.1f} GB/s  {k}")
