#!/bin/bash
# Every randomised sweep of tools/ in one pass with fresh seeds (run through gpurun from the repo root); prints each tool's verdict.
#   bash tools/fuzz_all.sh [seed offset]
S=${1:-10}
for cmd in "fuzz_ortho.py 300 $((S+1))" "fuzz_kernels.py 300 $((S+2))" "fuzz_drivers.py 60 $((S+3))" "fuzz_run_ahead.py 30 $((S+4))" \
           "fuzz_degenerate.py 400 $((S+5))" "fuzz_parity.py 30 $((S+6))" "fuzz_parity_lr.py 16 $((S+7))" "fuzz_multirank.py 10 $((S+8))" \
           "fuzz_spmm_sharded.py 6 $((S+9))" "stress_threads.py 8 4 $((S+10))" "fuzz_pending_basis.py 200 $((S+11))" "fuzz_pending_basis.py 150 $((S+12)) -1 wide" "fuzz_pending_basis.py 150 $((S+13)) -1 mixed" \
           "fuzz_degenerate_drivers.py 60 $((S+13))"; do
  echo "== $cmd"
  timeout -k 10 400 python3 tools/$cmd 2>&1 | grep -v "ortho_cd failed\|amdgpu.ids\|^ok " | tail -3 | cut -c1-300
done
