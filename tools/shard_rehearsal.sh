#!/bin/bash
# Rehearsal of an N-GPU strong-scaling run on ONE GPU: N ranks share the device, every rank holds n/N rows, the small
# products travel through the one-shot peer-to-peer all-reduce (the transport of the real run).  What it shows: the
# per-solve time of one shard including every cross-rank exchange -- the latency floor of the N-GPU run -- NOT its
# bandwidth (the ranks share one HBM).   bash tools/shard_rehearsal.sh [ranks...]
# (At most 5 ranks: the GPU pool's process guard ends a run with more than 6 processes on one card -- the ranks and their launcher --
#  so the 8 mailboxes of the driver's 8-GPU run cannot be rehearsed on one device here.)
set -e
mkdir -p gpurun_out
for np in "${@:-2 4}"; do
  for np1 in $np; do
    DIAGLIB_BENCH_SHARE_GPU=1 DIAGLIB_BENCH_NOPROFILE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $np1 \
      --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus $np1 --steps 10 --warmup 3 --no-cpu-baseline --no-random-leg \
      > gpurun_out/shard_np${np1}.json 2> gpurun_out/shard_np${np1}.err || { tail -5 gpurun_out/shard_np${np1}.err; exit 1; }
    echo "ranks $np1: $(cat gpurun_out/shard_np${np1}.json)"
  done
done
