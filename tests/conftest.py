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


@pytest.fixture()
def rng():
    return np.random.default_rng(12345)
