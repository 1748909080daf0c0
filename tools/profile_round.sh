#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02 [extra bench.py arguments, e.g. --solver lobpcg --n 10000000 --roots 32]
# Writes everything under gpurun_out/profile_<tag>/; the summaries are then copied into profiles/<tag>/.
# rocprofv3 always gets `python3 bench.py` directly after `--` and PMC passes carry only --kernel-trace.
set -e
TAG=${1:-r02}
shift || true
EXTRA="$@"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="python3 bench.py $EXTRA"
$B > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o run -- $B --steps 5 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/kt.log 2>&1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -o run -- $B --steps 1 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/mfma.log 2>&1
echo "mfma done"
KEY=$(python3 -c "import json,sys; print(json.load(open('$OUT/bench_default.json'))['config']['workload_key'])")
cp profiles/pmc_traffic.json $OUT/pmc_traffic.json 2>/dev/null || true
python3 tools/pmc_traffic.py $OUT/fetch/run_counter_collection.csv $OUT/write/run_counter_collection.csv $OUT/pmc_traffic.json "$KEY" > $OUT/pmc_traffic.txt
python3 tools/pmc_mfma.py $OUT/mfma/run_counter_collection.csv $OUT/mfma/run_kernel_trace.csv $OUT/mfma_util.json > $OUT/mfma_util.txt
python3 tools/kt_gaps.py $OUT/kt/run_kernel_trace.csv 2 > $OUT/kt_gaps.txt || true
cat $OUT/mfma_util.txt
ls $OUT/kt
