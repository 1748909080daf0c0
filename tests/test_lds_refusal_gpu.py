"""GPU: a refused request for more than 64 KiB of dynamic LDS is not swallowed: the launch is not
made, the engine drops to its 64 KiB shapes and redoes the operation.  Runs in a process of its own because the
refusal is forced through $DIAGLIB_AMD_FORCE_LDS_REFUSAL at engine creation and lowers the engine's limit for good."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from diaglib_amd import capi
from oracle.pyoracle import Oracle
o = Oracle(); ctx = capi.Context()
rng = np.random.default_rng(3)
n, l, k = 4096, 180, 13
x = np.asfortranarray(rng.standard_normal((n, l))); u = np.asfortranarray(rng.standard_normal((n, k)))
px, pu = ctx.panel(x), ctx.panel(u)
got = ctx.gram(px, pu)                       # 12-tile LDS-staged pass asks for ~117 KiB: refused once, redone narrower
want = o.gemm_tn(x, u)
assert np.abs(got - want).max() <= 64 * np.finfo(float).eps * (np.abs(x).T @ np.abs(u)).max(), np.abs(got - want).max()
# fused update + Gram and the Ritz sweep under the 64 KiB limit
q, _ = np.linalg.qr(x); pq = ctx.panel(np.asfortranarray(q))
ctx.ortho_vs_x(pq, pu)
un = pu.download()
assert np.abs(q.T @ un).max() < 1e-13 and np.abs(un.T @ un - np.eye(k)).max() < 1e-13
y = np.asfortranarray(rng.standard_normal((l, k)))
ev, r = ctx.panel(n, k), ctx.panel(n, k)
rn = ctx.ritz_residual(px, pq, y, np.arange(1.0, k + 1), k, np.zeros(k, np.int32), ev, r)
assert np.abs(ev.download() - x @ y).max() < 1e-10
assert np.abs(r.download() - (q @ y - (x @ y) * np.arange(1.0, k + 1))).max() < 1e-9
print("refusal handled")
"""


def test_refused_lds_request_is_redone_under_the_limit():
    env = dict(os.environ, DIAGLIB_AMD_FORCE_LDS_REFUSAL="1")
    p = subprocess.run([sys.executable, "-c", CODE % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "refusal handled" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


CODE_CHAIN = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from diaglib_amd import capi
from oracle.pyoracle import Oracle
o = Oracle(); ctx = capi.Context()
rng = np.random.default_rng(5)
n, l, k = 4096, 180, 13
q = np.asfortranarray(np.linalg.qr(rng.standard_normal((n, l)))[0])
for trial in range(2):
    u = np.asfortranarray(rng.standard_normal((n, k)))
    big = ctx.panel(np.asfortranarray(np.hstack([q, u])))
    px, pu = big.col(0, l), big.col(l, k)          # the drivers' layout: the device-driven chain takes this call
    ctx.ortho_vs_x(px, pu)                         # trial 0: the chain's first wide sweep asks for ~117 KiB and is refused
    got = pu.download()
    want = o.ortho_vs_x(q, u)[0]
    assert np.abs(got - want).max() < 1e-11, np.abs(got - want).max()
    assert np.abs(q.T @ got).max() < 1e-13 and np.abs(got.T @ got - np.eye(k)).max() < 1e-13
    assert np.array_equal(px.download(), q)
print("refusal inside a chain handled")
"""


def test_refused_lds_request_inside_a_device_driven_chain():
    '''the refusal must not surface half way through a chain (earlier speculative launches may already have
    updated U in place): the chain's launch paths are walked without launching first, a refusal there hands the call to
    the host-driven loop under the 64 KiB limit.'''
    env = dict(os.environ, DIAGLIB_AMD_FORCE_LDS_REFUSAL="1")
    p = subprocess.run([sys.executable, "-c", CODE_CHAIN % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "refusal inside a chain handled" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])


CODE_WIDE_RITZ = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from diaglib_amd import capi
ctx = capi.Context()
rng = np.random.default_rng(11)
n, l, m, k2 = 4096, 111, 37, 37                   # [Y | C2] = 74 columns = five tiles: ritz_kernel<5, ...> holds 48 KiB of static LDS
v = np.asfortranarray(rng.standard_normal((n, l))); av = np.asfortranarray(rng.standard_normal((n, l)))
pv, pav = ctx.panel(v), ctx.panel(av)
y = np.asfortranarray(rng.standard_normal((l, m))); c2 = np.asfortranarray(rng.standard_normal((l, k2)))
theta = np.arange(1.0, m + 1)
for trial in range(2):
    # trial 0: the wide sweep's request (48 KiB static + 71 KiB dynamic) is refused -- the engine's limit drops to 64 KiB, under
    # which that kernel can no longer be launched at all (its static part alone leaves 15 KiB): the call must come back
    # correct through the separate sweeps, and so must every later one
    ev, r, p, ap = ctx.panel(n, m), ctx.panel(n, m), ctx.panel(n, k2), ctx.panel(n, k2)
    rn = ctx.ritz_residual_p(pv, pav, y, theta, m, np.zeros(m, np.int32), ev, r, None, c2, p, ap)
    want_r = av @ y - (v @ y) * theta
    assert np.abs(ev.download() - v @ y).max() < 1e-10
    assert np.abs(r.download() - want_r).max() < 1e-9
    assert np.abs(p.download() - v @ c2).max() < 1e-10 and np.abs(ap.download() - av @ c2).max() < 1e-10
    assert np.allclose(rn[1], np.abs(want_r).max(0), rtol=1e-12)
print("refusal of the wide ritz sweep handled")
"""


def test_refused_lds_request_of_the_five_tile_ritz_sweep():
    '''the four- and five-tile Ritz kernels hold 48 KiB of static LDS; after a refusal (limit 64 KiB) a launch with
    48 KiB static + up to 64 KiB dynamic must not be attempted -- static and dynamic LDS are counted together.'''
    env = dict(os.environ, DIAGLIB_AMD_FORCE_LDS_REFUSAL="1")
    p = subprocess.run([sys.executable, "-c", CODE_WIDE_RITZ % ROOT], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0 and "refusal of the wide ritz sweep handled" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
