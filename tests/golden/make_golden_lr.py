#!/usr/bin/env python3
"""Golden fixtures for the linear-response driver, from the UNMODIFIED reference (caslr_eff_driver,
reference diaglib.f90:1024-1481; oracle/_ref built by `make -C oracle ref`).

    python tests/golden/make_golden_lr.py      ->  tests/golden/reference_lr_fixtures.npz

Data only: the problem is the portable one of oracle/oracle_ops.c (orc_lr_setup), the guesses are unit
vectors or the seeded array stored here, the outputs are what the reference returned (eigenvalues,
eigenvectors, ok) plus its verbose convergence table parsed into arrays.  dense_w holds the positive
eigenvalues of the full 2n x 2n pencil from scipy.linalg.eigh, as an independent check of the problem.
"""
import json
import os
import subprocess
import sys
import tempfile

os.environ.setdefault("OMP_NUM_THREADS", "8")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import numpy as np  # noqa: E402
from make_golden import parse_trace  # noqa: E402

CASES = [
    dict(name="lr_n300_unit", n=300, n_targ=4, n_max=8, max_iter=100, tol=1e-8, max_dav=20, guess="unit", seed=0),
    dict(name="lr_n300_rand", n=300, n_targ=4, n_max=8, max_iter=200, tol=1e-11, max_dav=10, guess="rand", seed=3),
    dict(name="lr_n500_rand", n=500, n_targ=6, n_max=11, max_iter=200, tol=1e-9, max_dav=20, guess="rand", seed=8),
    # the traditional driver (caslr_driver, i_alg = 0) with the harness' preconditioner for it (lrprec_1)
    dict(name="lrt_n300_unit", driver="caslr", n=300, n_targ=4, n_max=8, max_iter=100, tol=1e-8, max_dav=20, guess="unit", seed=0),
    dict(name="lrt_n300_rand", driver="caslr", n=300, n_targ=4, n_max=8, max_iter=300, tol=1e-10, max_dav=10, guess="rand", seed=3),
]

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, Reference
spec = json.loads(%r)
o = Oracle(); r = Reference()
n = spec['n']
o.lr_setup(n)
trad = spec.get('driver') == 'caslr'
fn = [o.fn(k) for k in ("orc_lr_apb", "orc_lr_amb", "orc_lr_spd", "orc_lr_smd", "orc_lr_prec1" if trad else "orc_lr_prec")]
g = np.load(spec['guess_file'])
solve = r.caslr if trad else r.caslr_eff
e, v, ok = solve(n, spec['n_targ'], spec['n_max'], spec['max_iter'], spec['tol'], spec['max_dav'], *fn, g, verbose=True)
sys.stdout.flush()
np.savez(spec['out'], eig=e, evec=v[:, :spec['n_targ']], ok=ok)
"""


def guess_array(kind, n, m, seed):
    if kind == "unit":
        g = np.zeros((2 * n, m), order="F")
        g[np.arange(m), np.arange(m)] = 1.0
        return g
    return np.asfortranarray(np.random.default_rng(seed).random((2 * n, m)) - 0.5)


def main():
    import scipy.linalg as sl
    from oracle.pyoracle import Oracle
    o = Oracle()
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for c in CASES:
            n, t, m = c["n"], c["n_targ"], c["n_max"]
            g = guess_array(c["guess"], n, m, c["seed"])
            gfile = os.path.join(tmp, "g.npy"); np.save(gfile, g)
            spec = dict(c, guess_file=gfile, out=os.path.join(tmp, "o.npz"))
            p = subprocess.run([sys.executable, "-c", CHILD % (ROOT, json.dumps(spec))], capture_output=True, text=True)
            if p.returncode != 0:
                raise RuntimeError(p.stderr)
            res = np.load(spec["out"])
            tr = parse_trace(p.stdout, t)
            apb, amb, spd, smd = o.lr_setup(n)
            a, b, s, d = 0.5 * (apb + amb), 0.5 * (apb - amb), 0.5 * (spd + smd), 0.5 * (spd - smd)
            big = np.block([[a, b], [b, a]]); met = np.block([[s, d], [-d, -s]])
            w = sl.eigh(met, big, eigvals_only=True)
            k = c["name"]
            out[k + "_spec"] = np.array(json.dumps(c))
            if c["guess"] != "unit":
                out[k + "_guess"] = g
            out[k + "_eig"] = res["eig"][:t]
            out[k + "_evec"] = res["evec"]
            out[k + "_ok"] = res["ok"]
            out[k + "_dense_w"] = np.sort(1.0 / w[w > 0])[:t]
            out[k + "_tr_iters"] = np.array(tr["iters"])
            out[k + "_tr_eig"] = tr["eig"]
            out[k + "_tr_rms"] = tr["rms"]
            out[k + "_tr_restarts"] = np.array(tr["restarts"])
            print(k, "iters", tr["iters"], "restarts", tr["restarts"], "ok", bool(res["ok"]), res["eig"][:t],
                  "dense diff", np.abs(res["eig"][:t] - out[k + "_dense_w"]).max())
    path = os.path.join(HERE, "reference_lr_fixtures.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
