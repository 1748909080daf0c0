"""Test helper: builds tests/_build/libdiaglib_hostsim.so = the PRODUCT's host logic + Fortran drivers
linked against oracle/hostsim_engine.cpp (host-memory engine running the oracle's C kernels) instead
of the HIP engine.  Test infrastructure only -- see the header of oracle/hostsim_engine.cpp."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tests", "_build")
# $DIAGLIB_HOSTSIM_SANITIZE=1 (tools/sanitize_cpu.sh): the same library with AddressSanitizer + UndefinedBehaviorSanitizer in every
# C / C++ translation unit -- the product's host logic and small dense kernels (2 650 lines of hand-written LAPACK replacements and
# pointer arithmetic) under a memory-error checker.  The caller preloads libasan / libubsan (python itself is not instrumented).
SANITIZE = bool(os.environ.get("DIAGLIB_HOSTSIM_SANITIZE"))
if SANITIZE:
    BUILD = os.path.join(ROOT, "tests", "_build_asan")
LIB = os.path.join(BUILD, "libdiaglib_hostsim.so")
FLANG = "/opt/rocm/lib/llvm/bin/flang"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"] if SANITIZE else []


def build() -> str:
    os.makedirs(BUILD, exist_ok=True)
    srcs_cxx = [os.path.join(ROOT, "diaglib_amd", "csrc", "host_logic.cpp"),
                os.path.join(ROOT, "diaglib_amd", "csrc", "smalldense.cpp"),
                os.path.join(ROOT, "oracle", "hostsim_engine.cpp")]
    srcs_c = [os.path.join(ROOT, "oracle", "oracle.c"), os.path.join(ROOT, "oracle", "oracle_ops.c")]
    srcs_f = [os.path.join(ROOT, "diaglib_amd", "fortran", f) for f in
              ("real_precision.f90", "diaglib.f90", "diaglib_cbind.f90")]
    deps = srcs_cxx + srcs_c + srcs_f + [os.path.join(ROOT, "diaglib_amd", "csrc", "dla_internal.h"),
                                         os.path.join(ROOT, "include", "diaglib_amd.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB

    def run(cmd):
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + p.stdout + p.stderr)

    objs = []
    for s in srcs_cxx:
        o = os.path.join(BUILD, os.path.basename(s) + ".o")
        run(["g++", "-O1" if SANITIZE else "-O2", "-march=x86-64-v3", "-std=c++17", "-fPIC", "-DSD_SINGLE_ISA", "-Wno-psabi"] + SAN + ["-c", s, "-o", o]); objs.append(o)
    for s in srcs_c:
        o = os.path.join(BUILD, os.path.basename(s) + ".o")
        run(["gcc", "-O1" if SANITIZE else "-O3", "-march=x86-64-v3", "-fopenmp", "-fPIC"] + SAN + ["-c", s, "-o", o]); objs.append(o)
    for s in srcs_f:
        o = os.path.join(BUILD, os.path.basename(s) + ".o")
        run([FLANG, "-O2", "-fPIC", "-c", s, "-o", o, "-module-dir", BUILD, "-I", BUILD]); objs.append(o)
    rt = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/*/libflang_rt.runtime.a")
    rt = rt or glob.glob(os.path.realpath("/opt/rocm") + "/lib/llvm/lib/clang/*/lib/*/libflang_rt.runtime.a")
    run(["g++", "-shared", "-fPIC", "-fopenmp"] + SAN + ["-o", LIB] + objs +
        ["-L" + os.path.dirname(rt[0]), "-lflang_rt.runtime", "-Wl,-Bsymbolic", "-lm"])
    return LIB
