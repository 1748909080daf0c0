"""pytest configuration: registers the `gpu` marker and shared fixtures.

`-m "not gpu"` tests run in the build container (no GPU): oracle vs golden fixtures, host
logic, C-ABI symbol table.  `-m gpu` tests are the parity tests proper: they drive the HIP
path through the C-ABI and compare with the oracle (oracle/) and the golden fixtures.
"""
import os
import sys

# the oracle's OpenMP loops are tiny in the tests: a 256-core GPU box must not spawn 256 threads
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("MKL_NUM_THREADS", "8")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
    if os.environ.get("DIAGLIB_HOSTSIM_SANITIZE"):
        # tools/sanitize_cpu.sh: every CPU test of the host-size kernels runs on the ASan + UBSan build (tests/hostsim.py), which
        # exports the same C-ABI -- tests/test_host_dense.py then exercises the instrumented smalldense.cpp
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostsim
        from diaglib_amd import capi
        capi.load(hostsim.build())


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx():
    """The HIP context.  Fails loudly when the extension or the GPU is missing."""
    from diaglib_amd import capi
    c = capi.Context()
    assert c.backend.startswith("hip:"), c.backend
    return c


# every option of the session context and its default (include/diaglib_amd.h)
_CTX_DEFAULTS = {1: 0, 2: 0, 3: 0, 4: 0, 5: 1, 6: 10, 7: 0, 8: 0, 9: 5000, 10: 1, 11: 1}


@pytest.fixture(autouse=True)
def _clean_ctx(request):
    """The HIP context is shared by the whole session; a test that leaves a shard, an option, a tune knob or a reduction hook behind
    silently changes the semantics of every later test (round 4: a forgotten shard turned rms norms into sums).  After every
    test that used it the context is put back and checked against its defaults."""
    yield
    if "ctx" not in request.fixturenames:
        return
    c = request.getfixturevalue("ctx")
    c.set_allreduce_hook(None, 1, 0)
    c.set_shard(-1, 0)
    for opt, val in _CTX_DEFAULTS.items():
        c.set_option(opt, val)
    for knob in range(8):
        c.set_option(100 + knob, 0)
    for opt, val in _CTX_DEFAULTS.items():
        assert c.get_option(opt) == val, (opt, c.get_option(opt))
    assert all(c.get_option(100 + knob) == 0 for knob in range(8))
    assert c.comm_info() == (1, 0), c.comm_info()


@pytest.fixture()
def rng():
    return np.random.default_rng(12345)
