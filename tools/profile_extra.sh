#!/bin/bash
# Side profiles of a round (run through gpurun from the repo root, after tools/profile_round.sh):
#   bash tools/profile_extra.sh r02
# the two wide-block BASELINE shapes on one GPU (bench line + rocprofv3 kernel stats for the LOBPCG one) and the kernel
# trace of one 250k-row shard.  Everything lands in gpurun_out/profile_<tag>/.
set -e
TAG=${1:-r02}
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
C5="--solver lobpcg --n 10000000 --roots 32 --tol 1e-12 --steps 3 --warmup 1 --no-cpu-baseline --no-random-leg"
python3 bench.py $C5 > $OUT/bench_lobpcg_cfg5shape_1gpu.json 2> $OUT/cfg5.err
echo "cfg5 bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt5 -o run -- python3 bench.py $C5 > $OUT/kt5.log 2>&1
cp $OUT/kt5/run_kernel_stats.csv $OUT/kernel_stats_lobpcg_cfg5shape.csv
rm -rf $OUT/kt5
echo "cfg5 stats done"
python3 bench.py --solver davidson --n 2000000 --roots 16 --steps 5 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/bench_davidson_cfg4shape_1gpu.json 2> $OUT/cfg4.err
python3 bench.py --solver lobpcg --steps 5 --warmup 1 --no-cpu-baseline --no-random-leg > $OUT/bench_lobpcg_cfg3.json 2> $OUT/cfg3.err
echo "cfg4 bench done"
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt250 -o run -- python3 bench.py --n 250000 --steps 4 --warmup 2 --no-cpu-baseline --no-random-leg > $OUT/kt250.log 2>&1
python3 tools/kt_gaps.py $OUT/kt250/run_kernel_trace.csv 3 --timeline > $OUT/kernel_trace_one_solve_250k_rows.txt
rm -rf $OUT/kt250
python3 bench.py --n 250000 --steps 10 --warmup 3 --no-cpu-baseline --no-random-leg > $OUT/bench_250k_rows.json 2>/dev/null
# BASELINE configs[1]: n = 5e5, 8 roots, Davidson, one GPU
python3 bench.py --n 500000 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_davidson_cfg2_500k_rows.json 2>/dev/null
echo "250k done"
