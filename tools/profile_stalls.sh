#!/bin/bash
# Stall counters of the sweeps of one workload, one rocprofv3 --pmc pass per counter group (--pmc is never combined with the
# API / memory-copy trace domains), summarised per kernel by tools/pmc_stalls.py:
#   bash tools/profile_stalls.sh <tag> [bench.py arguments]
# e.g.  bash tools/profile_stalls.sh cfg5 --solver lobpcg --n 10000000 --roots 32 --tol 1e-12
set -e
TAG=$1; shift
OUT=gpurun_out/stalls_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-random-leg $*"
i=0
for grp in "MfmaUtil LdsBankConflict" "VmemLatency LdsLatency" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
           "MemUnitStalled OccupancyPercent" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS"; do
  i=$((i+1))
  DIAGLIB_BENCH_NOPROFILE=1 timeout -k 10 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -o run -- python3 bench.py $ARGS > $OUT/p$i.log 2>&1 || { echo "pass $i ($grp) failed"; tail -3 $OUT/p$i.log; }
  echo "pass $i done: $grp"
done
python3 tools/pmc_stalls.py $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 $OUT/p5 > gpurun_out/stall_counters_$TAG.txt
find $OUT -name "*.csv" -size +20M -delete
head -40 gpurun_out/stall_counters_$TAG.txt
