"""diaglib_amd -- MI355X-native block eigensolver hot path behind diaglib's interface.

Layout (only what the path needs):
  csrc/     HIP kernels + engine (hip_engine.hip), host logic of the C-ABI (host_logic.cpp),
            host-size dense kernels (smalldense.cpp)
  fortran/  module diaglib (davidson_driver, lobpcg_driver, ortho_*) calling the C-ABI
  capi.py   ctypes mirror used by tests and bench.py
  _build.py in-tree build recipe (hipcc + flang, gfx950)
"""
from . import capi  # noqa: F401
from ._build import build  # noqa: F401

__all__ = ["capi", "build"]
