# latency rehearsal: one shard of an 8/4/2-GPU run on one GPU, with host-time accounting
set -e
mkdir -p gpurun_out
for n in 250000 2000000; do
  for np in 0 1; do
  if [ $np = 1 ]; then export DIAGLIB_BENCH_NOPROFILE=1; else unset DIAGLIB_BENCH_NOPROFILE; fi
  timeout -k 10 200 python bench.py --n $n --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/lat_np${np}_$n.json 2> gpurun_out/lat_np${np}_$n.err || true
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/lat_np${np}_$n.json || tail -3 gpurun_out/lat_np${np}_$n.err
  done
done
