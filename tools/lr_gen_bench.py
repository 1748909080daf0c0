#!/usr/bin/env python3
"""Device-resident linear-response and generalised solves at the benchmark's size (n = 2e6): time per solve and residuals, one
JSON line per driver (profiles/r03/bench_lr_gen_2e6.jsonl).  Operators = the library's sample operators (dla_synth_apbmul ...
dla_synth_metric, include/diaglib_amd.h), device callbacks, device-resident eigenvector block.

    python tools/lr_gen_bench.py [--n 2000000] [--steps 9]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diaglib_amd import capi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2_000_000)
    ap.add_argument("--steps", type=int, default=9)
    ap.add_argument("--ab-run-ahead", action="store_true", help="every driver with DLA_OPT_RUN_AHEAD 1 / 0 alternately, three times each")
    args = ap.parse_args()
    n = args.n
    ctx = capi.Context()
    ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 1)
    ctx.synth_setup(n, 0, n)
    f = capi.fn_address
    cases = [
        ("caslr_eff_driver", 4, 8, 1e-9, lambda ev, t, m, tol: ctx.caslr_eff_driver(n, t, m, 200, tol, 10, f("dla_synth_apbmul"), f("dla_synth_ambmul"),
                                                                                   f("dla_synth_spdmul"), f("dla_synth_smdmul"), f("dla_synth_lrprec2"), ev), 2),
        ("caslr_driver", 4, 8, 1e-9, lambda ev, t, m, tol: ctx.caslr_driver(n, t, m, 200, tol, 10, f("dla_synth_apbmul"), f("dla_synth_ambmul"),
                                                                           f("dla_synth_spdmul"), f("dla_synth_smdmul"), f("dla_synth_lrprec1"), ev), 2),
        ("gen_david_driver", 8, 13, 2e-13, lambda ev, t, m, tol: ctx.gen_david_driver(n, t, m, 200, tol, 20, 0.0, f("dla_synth_matvec"), f("dla_synth_precnd"),
                                                                                    f("dla_synth_metric"), ev), 1),
        ("lobpcg_driver(gen_eig)", 8, 13, 2e-13, lambda ev, t, m, tol: ctx.lobpcg_driver(n, t, m, 200, tol, 0.0, f("dla_synth_matvec"), f("dla_synth_precnd"), ev,
                                                                                       bvec=f("dla_synth_metric")), 1),
    ]
    for name, t, m, tol, solve, height in cases:
        rows = height * n
        guess = ctx.panel(rows, m).zero()
        top = np.eye(m, order="F")
        for j in range(m):
            ctx._chk(ctx.lib.dla_upload(ctx.h, guess.ptr + 8 * rows * j, top[:, j].ctypes.data, 8 * m))
        ev = ctx.panel(rows, m)

        def run():
            ctx.lib.dla_copy(ctx.h, ev.ptr, guess.ptr, 8 * rows * m)
            return solve(ev, t, m, tol)
        if args.ab_run_ahead:
            for rep in range(6):
                ctx.set_option(capi.OPT_RUN_AHEAD, 1 - rep % 2)
                run()
                ctx.sync(); ctx.reset_stats()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    eig, _, ok, info = run()
                ctx.sync()
                print(f"{name}: run-ahead {1 - rep % 2}: {(time.perf_counter() - t0) / args.steps * 1e3:.3f} ms per solve, "
                      f"{ctx.stats()['host_syncs'] / args.steps:.0f} host waits, {info['iters']} iterations", flush=True)
            ctx.set_option(capi.OPT_RUN_AHEAD, 1)
        run()
        ctx.sync(); ctx.reset_stats()
        per = []
        for _ in range(args.steps):
            t0 = time.perf_counter()
            eig, _, ok, info = run()
            ctx.sync()
            per.append(time.perf_counter() - t0)
        dt = float(np.median(per))          # (a solve is 15-25 ms with 20-50 host waits: one descheduled wait moves a mean of five)
        st = ctx.stats()
        gb = sum(st[c]["alg_bytes"] for c in ("gram", "gemm", "trmm", "ritz", "elem", "matvec", "precnd")) / args.steps / 1e9
        print(json.dumps({"driver": name, "n": n, "roots": t, "n_max": m, "tol": tol, "ms_per_solve": round(dt * 1e3, 3), "ms_min_max": [round(min(per) * 1e3, 3), round(max(per) * 1e3, 3)], "converged": bool(ok),
                          "iters": info["iters"], "restarts": info["restarts"], "eig": [round(float(e), 9) for e in eig[:t]],
                          "alg_GB_per_solve": round(gb, 2), "alg_TBps_over_wall": round(gb / dt / 1e3, 3),
                          "host_syncs_per_solve": st["host_syncs"] / args.steps, "callbacks": "device (sample operators)"}), flush=True)
        guess.free(); ev.free(); ctx.trim()


if __name__ == "__main__":
    main()
