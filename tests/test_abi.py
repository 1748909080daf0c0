"""CPU: the C-ABI shared library loads and exports every symbol include/diaglib_amd.h declares;
the Fortran module procedures are there under the names a Fortran caller links against; without a
GPU the product refuses to run (no CPU fallback)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diaglib_amd.h")


@pytest.fixture(scope="module")
def lib():
    from diaglib_amd import _build, capi
    _build.build()
    return capi.load()


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = set(re.findall(r"\b(dla_[a-z0-9_]+)\s*\(", src))
    names -= {"dla_matvec_fn", "dla_precnd_fn", "dla_allreduce_fn"}
    return sorted(names)


def test_every_declared_symbol_is_exported(lib):
    from diaglib_amd import capi
    declared = _declared()
    assert len(declared) >= 45
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(capi.EXPORTS)) == declared, set(capi.EXPORTS) ^ set(declared)


def test_fortran_module_procedures_present():
    from diaglib_amd import capi
    out = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True).stdout
    for proc in ("davidson_driver", "lobpcg_driver", "ortho_cd", "ortho_vs_x", "b_ortho", "b_ortho_vs_x", "ortho",
                 "diaglib_amd_config"):
        assert f"_QMdiaglibP{proc}" in out, proc     # flang mangling of module diaglib (reference diaglib.f90:166-167)


def test_no_oracle_or_reference_code_in_product():
    """The product library must not link the oracle or the compiled reference."""
    from diaglib_amd import capi
    out = subprocess.run(["nm", "-D", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out and "ref_davidson" not in out
    ldd = subprocess.run(["ldd", capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in ldd and "diaglib_ref" not in ldd and "mkl" not in ldd


def test_fails_loudly_without_gpu():
    """In a process with no visible GPU dla_create must return DLA_ERR_NO_DEVICE (never a CPU path)."""
    code = (
        "import ctypes,sys; sys.path.insert(0, %r)\n"
        "from diaglib_amd import capi\n"
        "L = capi.load(); h = ctypes.c_void_p()\n"
        "st = L.dla_create(ctypes.byref(h), 0)\n"
        "print('status', st)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert "status 1" in p.stdout or "status 4" in p.stdout, (p.stdout, p.stderr)
