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


def test_reduction_hand_over_carries_sc1_in_the_code_object(tmp_path):
    """gram_reduce_kernel hands its level-2 rows to the last-arriving block with relaxed agent-scope atomic stores /
    loads and a relaxed ticket.  That is only a hand-over because gfx950 lowers them to write-through (sc1) stores and sc1 loads
    (MI355X_MICROARCH.md, "Valid forms"); a toolchain that lowered them differently would break it silently.  Look at the ISA
    the build produced: the kernel must carry sc1 on those stores and loads."""
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    obj = os.path.join(ROOT, "diaglib_amd", "_obj", "hip_engine.hip.o")
    if not (os.path.exists(obj) and os.path.exists(os.path.join(llvm, "llvm-objdump"))):
        pytest.skip("no object file / llvm tools here")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([os.path.join(llvm, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(llvm, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
    dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", co], capture_output=True, text=True, check=True).stdout
    for variant in ("ILb1E", "ILb0E"):
        body, on = [], False
        for ln in dis.splitlines():
            if ln.endswith(">:"):
                on = ("gram_reduce_kernel" + variant) in ln
                continue
            if on:
                body.append(ln)
        assert body, variant
        stores = [ln for ln in body if "global_store_dwordx2" in ln]
        loads = [ln for ln in body if "global_load_dwordx2" in ln and "sc1" in ln]
        atomics = [ln for ln in body if "global_atomic_add" in ln]
        assert any("sc1" in ln for ln in stores), variant               # the level-2 rows go out write-through
        assert len(loads) >= 32, (variant, len(loads))                   # ... and come back through sc1 loads (up to 32 groups)
        assert atomics, variant                                          # the tickets (agent-scope RMWs, executed at the L2 / fabric)
