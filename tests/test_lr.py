"""Linear-response drivers caslr_eff_driver (reference diaglib.f90:1024-1481) and caslr_driver (:558-1022) against the fixtures the
unmodified reference produced (tests/golden/make_golden_lr.py).

CPU: the product's Fortran driver + host logic on the host-memory test engine (tests/hostsim.py), in a
child process.  GPU: the same driver on the HIP engine, callbacks in the default HOST mode (blocks staged
through pinned memory), i.e. what an unmodified Fortran caller gets."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = os.path.join(ROOT, "tests", "golden", "reference_lr_fixtures.npz")
FIX_HP = os.path.join(ROOT, "tests", "golden", "reference_ortho_fixtures.npz")     # caslr_driver with i_alg = 1
CASES = ["lr_n300_unit", "lr_n300_rand", "lr_n500_rand", "lrt_n300_unit", "lrt_n300_rand", "lrhp_n300_unit", "lrhp_n300_rand"]


def _fixture(name):
    return np.load(FIX_HP if name.startswith("lrhp") else FIX)

WORKER = r"""
import os, sys, json
os.environ.setdefault("OMP_NUM_THREADS", "4")
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from diaglib_amd import capi
if {hostsim!r}:
    import hostsim
    capi.load(hostsim.build())
from oracle.pyoracle import Oracle
spec = json.loads({spec!r})
o = Oracle()
n, t, m = spec["n"], spec["n_targ"], spec["n_max"]
o.lr_setup(n)
trad = spec.get("driver") == "caslr"
fn = [o.fn(k) for k in ("orc_lr_apb", "orc_lr_amb", "orc_lr_spd", "orc_lr_smd", "orc_lr_prec1" if trad else "orc_lr_prec")]
g = np.load(spec["guess_file"])
ctx = capi.Context()
assert ctx.backend.startswith("hostsim" if {hostsim!r} else "hip:"), ctx.backend
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
# Helmich-Paris route of the traditional driver (reference harness switch i_alg = 1, diaglib.f90:675,805-860)
ctx.set_option(capi.OPT_CASLR_ALGORITHM, 1 if spec["name"].startswith("lrhp") else 0)
solve = ctx.caslr_driver if trad else ctx.caslr_eff_driver
eig, vec, ok, info = solve(n, t, m, spec["max_iter"], spec["tol"], spec["max_dav"], *fn, g)
np.savez(spec["out"], eig=eig, vec=vec, ok=ok, iters=info["iters"], restarts=info["restarts"])
"""


def _guess(fx, name, spec):
    n, m = spec["n"], spec["n_max"]
    if spec["guess"] == "unit":
        g = np.zeros((2 * n, m), order="F")
        g[np.arange(m), np.arange(m)] = 1.0
        return g
    return np.asfortranarray(fx[name + "_guess"])


def _solve(tmp_path, fx, name, hostsim):
    spec = json.loads(str(fx[name + "_spec"]))
    gfile = tmp_path / "guess.npy"
    np.save(gfile, _guess(fx, name, spec))
    spec.update(guess_file=str(gfile), out=str(tmp_path / "res.npz"))
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, spec=json.dumps(spec), hostsim=hostsim))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    return spec, np.load(spec["out"])


def _check(fx, name, spec, res):
    from oracle.pyoracle import Oracle
    n, t = spec["n"], spec["n_targ"]
    assert bool(res["ok"]) and bool(fx[name + "_ok"])
    assert np.allclose(res["eig"][:t], fx[name + "_eig"], rtol=1e-11, atol=0)
    assert np.allclose(res["eig"][:t], fx[name + "_dense_w"], rtol=1e-10, atol=0)
    # iteration count and restarts of the reference's own trace
    assert abs(int(res["iters"]) - int(fx[name + "_tr_iters"])) <= 1
    assert int(res["restarts"]) == int(fx[name + "_tr_restarts"])
    v, vr = res["vec"][:, :t], fx[name + "_evec"]
    sgn = np.sign((v * vr).sum(0))
    assert np.abs(v * sgn - vr).max() < 1e-6 * max(1.0, np.abs(vr).max())
    # residual of the generalised problem (A B; B A) x = w (S D; -D -S) x
    apb, amb, spd, smd = Oracle().lr_setup(n)
    a, b, s, d = 0.5 * (apb + amb), 0.5 * (apb - amb), 0.5 * (spd + smd), 0.5 * (spd - smd)
    big = np.block([[a, b], [b, a]]); met = np.block([[s, d], [-d, -s]])
    for i in range(t):
        x = v[:, i]
        r = big @ x - res["eig"][i] * (met @ x)
        assert np.linalg.norm(r) / np.linalg.norm(big @ x) < max(50 * spec["tol"], 1e-9)


@pytest.mark.parametrize("name", CASES)
def test_lr_driver_on_host_engine_matches_reference_fixture(tmp_path, name):
    fx = _fixture(name)
    spec, res = _solve(tmp_path, fx, name, True)
    _check(fx, name, spec, res)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_lr_driver_gpu_matches_reference_fixture(tmp_path, name):
    fx = _fixture(name)
    spec, res = _solve(tmp_path, fx, name, False)
    _check(fx, name, spec, res)
