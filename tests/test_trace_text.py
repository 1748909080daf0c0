"""The verbose convergence table is the reference's de-facto trace format (diaglib.f90:1671, 1752).
The product prints the same table: here its text is captured from a child process, parsed with the SAME regular
expression that parsed the reference's output when the fixtures were made (tests/golden/make_golden.py), and compared
iteration by iteration with the reference's recorded table on the cases whose history is robust to rounding
(unit-vector guesses on the reference's dense matrix).

CPU: Fortran drivers + host logic on the host-memory test engine.  GPU: the HIP engine, host callbacks."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
CASES = ["dav_n1000_unit", "dav_n2000_unit", "lob_n1000_unit", "lob_n2000_unit", "gdav_n600_unit", "glob_n600_unit"]

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
sp = json.loads({spec!r})
o = Oracle()
n, t, m = sp["n"], sp["n_targ"], sp["n_max"]
o.dense_setup(n)
mv, pc = o.fn("orc_dense_matvec"), o.fn("orc_dense_precnd")
bv = None
if sp.get("gen"):
    o.metric_setup(n); bv = o.fn("orc_metric_matvec")
g = np.zeros((n, m), order="F"); g[np.arange(m), np.arange(m)] = 1.0
ctx = capi.Context()
ctx.set_option(capi.OPT_CALLBACKS_ON_DEVICE, 0)
if sp["solver"] == "davidson":
    ctx.davidson_driver(n, t, m, sp["max_iter"], sp["tol"], sp["max_dav"], sp["shift"], mv, pc, g, verbose=True)
elif sp["solver"] == "gen_davidson":
    ctx.gen_david_driver(n, t, m, sp["max_iter"], sp["tol"], sp["max_dav"], sp["shift"], mv, pc, bv, g, verbose=True)
else:
    ctx.lobpcg_driver(n, t, m, sp["max_iter"], sp["tol"], sp["shift"], mv, pc, g, verbose=True, bvec=bv)
sys.stdout.flush()
"""


def _run(tmp_path, name, hostsim):
    from make_golden import parse_trace
    fx = np.load(os.path.join(ROOT, "tests", "golden", "reference_fixtures.npz"))
    spec = next(s for s in json.loads(str(fx["driver_specs"])) if s["name"] == name)
    script = tmp_path / "w.py"
    script.write_text(WORKER.format(root=ROOT, spec=json.dumps(spec), hostsim=hostsim))
    p = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    tr = parse_trace(p.stdout, spec["n_targ"])
    assert tr["iters"] == int(fx[name + "_tr_iters"]), (tr["iters"], int(fx[name + "_tr_iters"]))
    assert tr["restarts"] == int(fx[name + "_tr_restarts"])
    assert np.array_equal(tr["done"], fx[name + "_tr_done"])                     # same locking pattern
    assert np.array_equal(tr["n_act_added"], fx[name + "_tr_n_act_added"])       # same block sizes
    assert np.abs(tr["eig"] - fx[name + "_tr_eig"]).max() < 1e-9                  # 12 printed decimals
    big = fx[name + "_tr_rms"] > 1e-6                                             # well above rounding level
    assert np.allclose(tr["rms"][big], fx[name + "_tr_rms"][big], rtol=5e-3)
    assert np.allclose(tr["rmax"][big], fx[name + "_tr_rmax"][big], rtol=5e-3)
    # residuals at the converged level (1e-9 .. 1e-6) depend on the last bits of the Ritz vectors: same magnitude
    mid = (fx[name + "_tr_rms"] > 1e-9) & ~big
    assert np.all(tr["rms"][mid] < 2.0 * fx[name + "_tr_rms"][mid]) and np.all(tr["rms"][mid] > 0.5 * fx[name + "_tr_rms"][mid])
    # the header, the timing footer and (Davidson family only, like the reference) the block box are there
    assert "iterations (tol=" in p.stdout and "timings for" in p.stdout
    assert ("# new vectors added:" in p.stdout) == (spec["solver"] != "lobpcg")


@pytest.mark.parametrize("name", CASES)
def test_verbose_table_matches_reference_on_host_engine(tmp_path, name):
    _run(tmp_path, name, True)


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_verbose_table_matches_reference_gpu(tmp_path, name):
    _run(tmp_path, name, False)
