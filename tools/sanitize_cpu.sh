#!/bin/bash
# CPU sanitizer job: the product's host side -- diaglib_amd/csrc/host_logic.cpp, smalldense.cpp -- and the
# host-memory test engine oracle/hostsim_engine.cpp (+ the oracle's C kernels underneath) compiled with
# -fsanitize=address,undefined, and every test of the `-m "not gpu"` suite that runs them (drivers, orthogonalisation loops,
# get_coeffs, the linear-response drivers, the 2- and 8-rank gloo runs, the trace comparisons) executed under it.
# CPU only: never on the GPU box (gpurun refuses GPU sanitizers).   bash tools/sanitize_cpu.sh [pytest args]
# Result: profiles/r06/sanitizer_cpu.txt
set -o pipefail
cd "$(dirname "$0")/.."
export DIAGLIB_HOSTSIM_SANITIZE=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
# (python, numpy, torch leak by design at exit: leak detection off; everything else aborts the run at the first report)
export ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:detect_stack_use_after_return=0"
export UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1"
out=profiles/r06/sanitizer_cpu.txt
mkdir -p profiles/r06
{
  echo "# $(date -u +%FT%TZ)  gcc $(gcc -dumpversion), -fsanitize=address,undefined -fno-sanitize-recover=undefined -O1 -g"
  echo "# instrumented: diaglib_amd/csrc/host_logic.cpp, diaglib_amd/csrc/smalldense.cpp, oracle/hostsim_engine.cpp, oracle/oracle.c, oracle/oracle_ops.c"
  echo "# ASAN_OPTIONS=$ASAN_OPTIONS  UBSAN_OPTIONS=$UBSAN_OPTIONS"
} > "$out"
python -m pytest tests/test_host_dense.py tests/test_hostsim.py tests/test_ortho_qr.py tests/test_lr.py tests/test_trace_text.py tests/test_spmm_sharded.py \
       -q -m "not gpu" -p no:cacheprovider "$@" 2>&1 | tee -a "$out"
rc=${PIPESTATUS[0]}
echo "# exit code $rc; sanitizer reports in the output above: $(grep -c 'ERROR: AddressSanitizer\|runtime error:' "$out")" >> "$out"
exit $rc
