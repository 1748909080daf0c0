#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01
# Writes everything under gpurun_out/profile_<tag>/; the summaries are then copied into profiles/<tag>/.
# rocprofv3 always gets `python3 bench.py` directly after `--` and PMC passes carry only --kernel-trace.
set -e
TAG=${1:-r01}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o run -- $B --steps 5 --warmup 1 --no-cpu-baseline > $OUT/kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline > $OUT/mfma.log 2>&1
echo "mfma done"
python3 tools/pmc_traffic.py $OUT/fetch/run_counter_collection.csv $OUT/write/run_counter_collection.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.txt
python3 tools/pmc_mfma.py $OUT/mfma/run_counter_collection.csv $OUT/mfma/run_kernel_trace.csv $OUT/mfma_util.json > $OUT/mfma_util.txt
cat $OUT/mfma_util.txt
ls $OUT/kt
