"""Build recipe of the native library (HIP engine + host logic + Fortran drivers).

Everything is compiled in-tree for gfx950 and linked into ``diaglib_amd/lib/libdiaglib_amd.so``
(git-ignored, travels to the GPU box with the snapshot).  Tool chain: hipcc (device + C++ host),
flang (Fortran drivers; its runtime is linked statically), RCCL from /opt/rocm.
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
OBJ = os.path.join(PKG, "_obj")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libdiaglib_amd.so")

ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
FLANG = os.path.join(ROCM, "lib", "llvm", "bin", "flang")
ARCH = "gfx950"

CXX_SOURCES = ["csrc/host_logic.cpp", "csrc/smalldense.cpp"]
WIDE_SIMD_SOURCES = ["csrc/smalldense.cpp"]
HIP_SOURCES = ["csrc/hip_engine.hip"]
F90_SOURCES = ["fortran/real_precision.f90", "fortran/diaglib.f90", "fortran/diaglib_cbind.f90"]  # order matters
HEADERS = ["csrc/dla_internal.h", os.path.join(ROOT, "include", "diaglib_amd.h")]


def _run(cmd: list[str], verbose: bool) -> None:
    if verbose:
        print("+", " ".join(cmd), flush=True)
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        sys.stderr.write(p.stdout + p.stderr)
        raise RuntimeError("build step failed: " + " ".join(cmd))


def _newer(src: list[str], dst: str) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in src if os.path.exists(s))


def _flang_rt_dir() -> str:
    hits = glob.glob(os.path.join(ROCM, "lib", "llvm", "lib", "clang", "*", "lib", "*", "libflang_rt.runtime.a"))
    if not hits:
        hits = glob.glob(os.path.join(os.path.realpath(ROCM), "lib", "llvm", "lib", "clang", "*", "lib", "*",
                                      "libflang_rt.runtime.a"))
    if not hits:
        raise RuntimeError("flang runtime (libflang_rt.runtime.a) not found under " + ROCM)
    return os.path.dirname(hits[0])


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile what is out of date and return the path of the shared library."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(PKG, h) for h in HEADERS]
    objs = []
    for s in HIP_SOURCES:
        src = os.path.join(PKG, s)
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        if force or _newer([src] + hdrs, o):
            _run([HIPCC, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-c", src, "-o", o],
                 verbose)
        objs.append(o)
    for s in CXX_SOURCES:
        src = os.path.join(PKG, s)
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        if force or _newer([src] + hdrs, o):
            # host-size dense kernels want AVX2/FMA (every x86 host of an MI355X node has them)
            _run([HIPCC, "-x", "c++", "-O3", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-c", src, "-o", o], verbose)
        objs.append(o)
        if s in WIDE_SIMD_SOURCES:
            # the same file once more for AVX-512 hosts (EPYC Zen 4 / 5), picked at run time (smalldense.cpp)
            o4 = os.path.join(OBJ, os.path.basename(s) + ".v4.o")
            if force or _newer([src] + hdrs, o4):
                _run([HIPCC, "-x", "c++", "-O3", "-march=x86-64-v4", "-mprefer-vector-width=512", "-std=c++17", "-fPIC",
                      "-DSD_NS=sd_v4", "-DSD_IMPL_ONLY", "-c", src, "-o", o4], verbose)
            objs.append(o4)
    f_objs = []
    rebuild_f = force
    for s in F90_SOURCES:
        src = os.path.join(PKG, s)
        o = os.path.join(OBJ, os.path.basename(s) + ".o")
        if rebuild_f or _newer([src], o):
            rebuild_f = True  # later modules depend on earlier ones
            _run([FLANG, "-O2", "-fPIC", "-c", src, "-o", o, "-module-dir", OBJ, "-I", OBJ], verbose)
        f_objs.append(o)
    objs += f_objs
    if force or _newer(objs, LIB):
        _run([HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs +
             ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-L" + _flang_rt_dir(), "-lflang_rt.runtime",
              "-Wl,-rpath," + os.path.join(ROCM, "lib"),
              # bind the module procedures (_QMdiaglibP...) inside the library: a caller process may also
              # hold another library that defines the same Fortran module name
              "-Wl,-Bsymbolic"], verbose)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
