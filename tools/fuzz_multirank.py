#!/usr/bin/env python3
"""Randomised multi-rank runs on ONE GPU (2 .. 4 ranks sharing the device: peer-to-peer mailboxes or the reduction hook over
gloo) against the single-rank run of the same problem AND the oracle (oracle/pyoracle.py, one rank): row counts that are no multiple of anything, block widths 2 .. 37,
Davidson / LOBPCG / generalised variants on the built-in operators.  Checked: every rank takes identical decisions (same
eigenvalues to the bit, same iteration count), the result equals the single-rank result, the shards stitch together.

    python tools/fuzz_multirank.py [cases] [seed]"""
import os
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_multirank_gpu as tm  # noqa: E402  (worker script and launcher of the test suite)
from oracle.pyoracle import Oracle  # noqa: E402  (the checker: every multi-rank result is also held against the oracle's)

ORACLE = Oracle()

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
bad = 0
for it in range(cases):
    rng = np.random.default_rng([seed, it])
    world = int(rng.integers(2, 5))
    transport = str(rng.choice(["p2p", "p2p", "hook"]))
    solver = str(rng.choice(["davidson", "davidson", "lobpcg", "gen_david", "gen_lobpcg"]))
    t = int(rng.choice([rng.integers(1, 9), rng.integers(9, 33)]))
    m = int(min(37, t + rng.integers(1, 8)))
    n = int(rng.integers(40_000, 300_000))
    spec = dict(n=n, n_targ=t, n_max=m, tol=float(rng.choice([1e-8, 1e-10])), solver=solver, guess="unit", transport=transport)
    with tempfile.TemporaryDirectory() as td:
        d1 = Path(td) / "w1"; d1.mkdir()
        dn = Path(td) / "wn"; dn.mkdir()
        try:
            one = tm._run_world(d1, spec, 1)[0]
            many = tm._run_world(dn, spec, world)
        except AssertionError as e:
            bad += 1; print("FAIL (a rank died)", dict(spec, world=world), str(e)[-400:], flush=True); continue
        res = {}
        res["all_ok"] = bool(one["ok"]) and all(bool(r["ok"]) for r in many)
        res["same_bits_on_all_ranks"] = all(np.array_equal(many[0]["eig"], r["eig"]) and int(r["iters"]) == int(many[0]["iters"]) for r in many)
        res["eig_vs_one_rank"] = float(np.abs(many[0]["eig"][:t] / one["eig"][:t] - 1.0).max())
        res["iters"] = (int(many[0]["iters"]), int(one["iters"]))
        v = np.vstack([r["vec"] for r in many]); v1 = one["vec"]
        sgn = np.sign((v1 * v).sum(0))
        res["vec"] = float(np.abs(v * sgn - v1)[:, :t].max())
        eo, vo, oko, ito = tm.oracle_run(ORACLE, spec)
        res["eig_vs_oracle"] = float(np.abs(many[0]["eig"][:t] / eo[:t] - 1.0).max())
        res["iters_oracle"] = ito
        good = (res["all_ok"] and res["same_bits_on_all_ranks"] and res["eig_vs_one_rank"] < 1e-10 and res["vec"] < 1e-5 and
                abs(res["iters"][0] - res["iters"][1]) <= max(1, res["iters"][1] // 5) and v.shape == v1.shape and
                oko and res["eig_vs_oracle"] < 1e-10 and abs(res["iters"][0] - ito) <= max(2, ito // 5))
        print(("ok  " if good else "FAIL"), dict(spec, world=world), res, flush=True)
        bad += 0 if good else 1
print(f"{cases} multi-rank cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
