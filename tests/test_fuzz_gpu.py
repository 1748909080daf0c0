"""A bounded, seeded slice of every randomised sweep in tools/ (tools/fuzz_all.sh runs them at length): the reference has no tests
at all (its strategy is main.f90:283-401, one printed comparison with LAPACK), so these sweeps are this build's own bar -- they found
three real defects in round 4 -- and the driver's `-m gpu` run exercises them.  Each tool exits non-zero on any failure."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SLICES = [
    ("fuzz_ortho.py", ["200", "101"]),                # ortho_cd / ortho_vs_x on random shapes against the oracle
    ("fuzz_kernels.py", ["200", "102"]),              # the block kernels on random shapes
    ("fuzz_drivers.py", ["30", "103"]),               # the four drivers against dense solutions
    ("fuzz_run_ahead.py", ["12", "104"]),             # run-ahead on / off: identical bits
    ("fuzz_degenerate.py", ["300", "105"]),           # duplicate / zero / dependent columns
    ("fuzz_parity.py", ["20", "106"]),                # drivers against the oracle and the unmodified reference (oracle/_ref), every case
                                                     # also without pending blocks and with both sweep schedules
    ("fuzz_parity_lr.py", ["12", "107"]),             # linear-response drivers against the reference
    ("fuzz_multirank.py", ["5", "108"]),             # 2 .. 4 ranks on one GPU against the single-rank run
    ("fuzz_spmm_sharded.py", ["3", "109"]),          # the sharded sparse sample operator
    ("fuzz_degenerate_drivers.py", ["21", "112"]),    # the drivers on operators with multiple / zero / all-equal eigenvalues: never ok with a wrong answer
    ("fuzz_pending_basis.py", ["60", "110"]),         # bases grown through dla_expand_project modes 4 / 5: (panel D) orthonormal, h exact
    ("fuzz_pending_basis.py", ["40", "111", "-1", "wide"]),   # ... blocks of 17 .. 40 columns (mode 4, host-driven loops beyond 223 columns)
    ("fuzz_pending_basis.py", ["40", "112", "-1", "mixed"]),  # ... mode 5 with a width per block (1 .. 40): wide blocks behind pending narrow ones, bases beyond the device copy of D
]


@pytest.mark.parametrize("tool,args", SLICES, ids=[t + ("-" + a[-1] if a[-1] == "wide" else "") for t, a in SLICES])
def test_fuzz_slice(tool, args):
    env = dict(os.environ, OMP_NUM_THREADS="4", MKL_NUM_THREADS="4", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    tail = (p.stdout[-1500:] + "\n" + p.stderr[-1500:])
    assert p.returncode == 0, tail
    assert "FAIL" not in p.stdout, tail
