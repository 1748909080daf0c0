#!/usr/bin/env python3
"""Golden fixtures from the UNMODIFIED reference (oracle/_ref built by `make -C oracle ref`) for
  * `ortho` (reference diaglib.f90:3052-3092, module symbol _QMdiaglibPortho): Householder QR + U R^-1, and
  * `caslr_driver` with the harness switch i_alg = 1 (Helmich-Paris reduced problem, :805-860).

    python tests/golden/make_golden_ortho.py      ->  tests/golden/reference_ortho_fixtures.npz

Data only: inputs are seeded arrays stored here, outputs are what the reference returned.
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
from make_golden_lr import guess_array  # noqa: E402

QR_CASES = [(257, 1, 1.0, 1), (257, 5, 1e3, 2), (300, 13, 1.0, 3), (300, 13, 1e8, 4), (400, 21, 1e12, 5), (64, 37, 1e2, 6)]
LR_CASES = [
    dict(name="lrhp_n300_unit", driver="caslr", n=300, n_targ=4, n_max=8, max_iter=100, tol=1e-8, max_dav=20, guess="unit", seed=0),
    dict(name="lrhp_n300_rand", driver="caslr", n=300, n_targ=4, n_max=8, max_iter=300, tol=1e-10, max_dav=10, guess="rand", seed=3),
]

CHILD = r"""
import sys, json, numpy as np
sys.path.insert(0, %r)
from oracle.pyoracle import Oracle, Reference
spec = json.loads(%r)
o = Oracle(); r = Reference()
r.set_i_alg(1)
n = spec['n']
o.lr_setup(n)
fn = [o.fn(k) for k in ("orc_lr_apb", "orc_lr_amb", "orc_lr_spd", "orc_lr_smd", "orc_lr_prec1")]
g = np.load(spec['guess_file'])
e, v, ok = r.caslr(n, spec['n_targ'], spec['n_max'], spec['max_iter'], spec['tol'], spec['max_dav'], *fn, g, verbose=True)
sys.stdout.flush()
np.savez(spec['out'], eig=e, evec=v[:, :spec['n_targ']], ok=ok)
"""


def main():
    import scipy.linalg as sl
    from oracle.pyoracle import Oracle, Reference
    ref, o = Reference(), Oracle()
    out = {"qr_count": len(QR_CASES)}
    # `ortho` works on the module-level LAPACK workspace (work, tau, lwork: reference diaglib.f90:155-161) that only exists
    # while a driver runs (allocated at :1600-1617, released at :1845).  So the calls are made from inside a driver run:
    # the first matvec callback of a small Davidson solve with n_max = 40 >= every k below.
    import ctypes as C
    c_dp, c_ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    nd, td, md = 500, 2, 40
    o.dense_setup(nd)
    dense_mv = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp)(("orc_dense_matvec", o.lib))
    state = {"done": False}

    def mv(pn, pm, px, pax):
        if not state["done"]:
            state["done"] = True
            for i, (n, k, cond, seed) in enumerate(QR_CASES):
                rng = np.random.default_rng(seed)
                q = np.linalg.qr(rng.standard_normal((n, k)))[0]
                sv = np.logspace(0, -np.log10(cond), k) if k > 1 else np.ones(1)
                u = np.asfortranarray((q * sv[None, :]) @ np.linalg.qr(rng.standard_normal((k, k)))[0])
                out[f"qr{i}_in"], out[f"qr{i}_out"], out[f"qr{i}_cond"] = u, ref.ortho(u), cond
        dense_mv(pn, pm, px, pax)

    cb = C.CFUNCTYPE(None, c_ip, c_ip, c_dp, c_dp)(mv)
    g0 = np.zeros((nd, md), order="F"); g0[np.arange(md), np.arange(md)] = 1.0
    ref.davidson(nd, td, md, 5, 1e-6, 10, 0.0, C.cast(cb, C.c_void_p).value, o.fn("orc_dense_precnd"), g0)
    assert state["done"]
    with tempfile.TemporaryDirectory() as tmp:
        for c in LR_CASES:
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
            name = c["name"]
            out[name + "_spec"] = np.array(json.dumps(c))
            out[name + "_eig"] = res["eig"][:t]
            out[name + "_evec"] = res["evec"]
            out[name + "_ok"] = bool(res["ok"])
            out[name + "_dense_w"] = np.sort(1.0 / w[w > 0])[:t]
            out[name + "_tr_iters"] = np.array(tr["iters"])
            out[name + "_tr_restarts"] = np.array(tr["restarts"])
            if c["guess"] != "unit":
                out[name + "_guess"] = g
            print(name, "ok" if res["ok"] else "NOT converged", "iters", tr["iters"], "eig", res["eig"][:t])
    np.savez_compressed(os.path.join(HERE, "reference_ortho_fixtures.npz"), **out)
    print("written", os.path.join(HERE, "reference_ortho_fixtures.npz"))


if __name__ == "__main__":
    main()
