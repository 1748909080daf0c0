#!/usr/bin/env python3
"""LOBPCG with and without the pending-factor form of the W block (dla_expand_project mode 3; $DIAGLIB_AMD_NO_PENDING switches
it off): iterations, time per solve, final residual -- each leg in its own process (the switch is read once).

    python tools/lobpcg_pending_ab.py [n] [roots] [tol]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n, roots, tol = (sys.argv + ["10000000", "32", "1e-12"])[1:4]
for rep in range(2):
    for off in (0, 1):
        env = dict(os.environ, DIAGLIB_BENCH_NOPROFILE="1")
        if off:
            env["DIAGLIB_AMD_NO_PENDING"] = "1"
        else:
            env.pop("DIAGLIB_AMD_NO_PENDING", None)
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--solver", "lobpcg", "--n", n, "--roots", roots, "--tol", tol,
                            "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-random-leg"], capture_output=True, text=True, env=env)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        d = json.loads(line[-1]) if line else {}
        print(f"n={n} roots={roots} tol={tol} pending factor {'off' if off else 'on '}: {d.get('ms_per_step')} ms, {d.get('iters')} iterations, "
              f"{d.get('host_syncs')} host waits / {d.get('steps')} solves", flush=True)
