#!/usr/bin/env python3
"""Randomised runs of the sharded sample sparse operator (dla_spmm_setup_csr_sharded): 2 .. 4 ranks on one GPU, hook and
peer-to-peer transports, half-bandwidths 1 .. 200 (ELLPACK widths 3 .. 401: the register-resident and the generic kernel),
block widths 2 .. 40 (one or several exchanges per product), row counts that are no multiple of anything -- checks of
tests/test_spmm_sharded.py (product against scipy on every shard, Davidson against the single-rank solve).

    python tools/fuzz_spmm_sharded.py [cases] [seed]"""
import os
import sys
import tempfile
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_spmm_sharded as ts  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = 0
for it in range(cases):
    rng = np.random.default_rng([seed, it])
    world = int(rng.integers(2, 5))
    hb = int(rng.choice([rng.integers(1, 8), rng.integers(8, 40), rng.integers(40, 200)]))
    m = int(rng.integers(2, 41)); t = int(rng.integers(1, m + 1))
    n = int(rng.integers(max(20_000, 64 * world + 4 * hb * world), 200_000))
    spec = dict(backend="hip", transport=str(rng.choice(["p2p", "p2p", "hook"])), n=n, n_targ=min(t, 12), n_max=m, half_band=hb, tol=1e-9)
    with tempfile.TemporaryDirectory() as td:
        try:
            ts._check(Path(td), spec, world)
            print("ok  ", dict(spec, world=world), flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", dict(spec, world=world), str(e)[:600].replace("\n", " | "), flush=True)
print(f"{cases} sharded-operator cases, {bad} failures", flush=True)
sys.exit(1 if bad else 0)
