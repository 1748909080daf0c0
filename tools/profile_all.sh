#!/bin/bash
# Everything profiles/<tag>/ holds, from the build in the tree, in ONE pass on one GPU box (run through gpurun from the repo root):
#   bash tools/profile_all.sh r03
# = tools/profile_round.sh (headline: bench line, rocprofv3 kernel statistics + one-solve trace, FETCH_SIZE / WRITE_SIZE / MFMA
# counter passes), the same counter passes on the cfg 5 shape, tools/profile_extra.sh (cfg 3 / 4 / 5 bench lines, 250 k-row
# shard), the shard rehearsal, the linear-response / generalised runs, the host-time report, the k x k step's time stamps, the
# host-mode probe and the HIP legs of the floor probes.  Results are collected in gpurun_out/profiles_<tag>/ (copy to profiles/<tag>/).
# A gpurun call is limited to 20 minutes: the pass can be cut in four -- bash tools/profile_all.sh r05 A | B | C | D (default: all).
set -e
TAG=${1:-r05}
PART=${2:-ABCD}
OUT=gpurun_out/profiles_$TAG
[[ $PART == *A* ]] && rm -rf $OUT
mkdir -p $OUT
P=gpurun_out/profile_$TAG
if [[ $PART == *A* ]]; then
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
cp $P/bench_default.json $P/pmc_traffic.txt $P/mfma_util.json $P/mfma_util.txt $OUT/
cp $P/kt/run_kernel_stats.csv $OUT/kernel_stats_bench_steps5.csv
cp $P/kt_gaps.txt $OUT/kernel_trace_one_solve.txt
cp $P/fetch/run_counter_collection.csv $OUT/pmc_fetch_counter_collection.csv
cp $P/write/run_counter_collection.csv $OUT/pmc_write_counter_collection.csv
cp $P/pmc_traffic.json $OUT/pmc_traffic_headline.json
echo "headline done"
bash tools/profile_round.sh ${TAG}c5 --solver lobpcg --n 10000000 --roots 32 --tol 1e-12 --no-cpu-baseline --no-random-leg > $OUT/profile_round_c5.log 2>&1
cp gpurun_out/profile_${TAG}c5/mfma_util.txt $OUT/mfma_util_cfg5shape.txt
cp gpurun_out/profile_${TAG}c5/mfma_util.json $OUT/mfma_util_cfg5shape.json
cp gpurun_out/profile_${TAG}c5/pmc_traffic.txt $OUT/pmc_traffic_cfg5shape.txt
cp gpurun_out/profile_${TAG}c5/pmc_traffic.json $OUT/pmc_traffic_cfg5shape.json
echo "cfg5 counters done"
bash tools/profile_extra.sh $TAG > $OUT/profile_extra.log 2>&1
cp $P/bench_lobpcg_cfg5shape_1gpu.json $P/bench_davidson_cfg4shape_1gpu.json $P/bench_lobpcg_cfg3.json $P/bench_250k_rows.json $P/bench_davidson_cfg2_500k_rows.json \
   $P/kernel_stats_lobpcg_cfg5shape.csv $P/kernel_trace_one_solve_250k_rows.txt $OUT/
echo "extra done"
fi
if [[ $PART == *B* ]]; then
bash tools/shard_rehearsal.sh 1 2 4 5 > $OUT/shard_rehearsal_2e6.txt 2>&1
python3 tools/lr_gen_bench.py > $OUT/bench_lr_gen_2e6.jsonl 2> $OUT/lr_gen.err
DIAGLIB_AMD_HOSTTIME=1 DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --n 250000 --steps 20 --warmup 3 --no-cpu-baseline --no-random-leg > $OUT/hosttime_250k_rows.txt 2>&1
DIAGLIB_AMD_CHAIN_DEBUG=1 DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-random-leg 2>&1 | grep -A4 "chain k=13 m=39" | tail -5 | cut -c1-260 > $OUT/chain_timing_k13_m39.txt
python3 tools/host_mode_probe.py 2>&1 | tail -8 > $OUT/host_mode_probe.txt
# Davidson at n_max = 21 (cfg 4 shape): blocks of up to 16 columns keep their closing passes pending (r06) / everything finished in memory
for r in 1 2 3; do
  DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --roots 16 --steps 10 --warmup 3 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('davidson n=2e6 16 roots n_max=21, pending blocks on :', d['ms_per_step'], 'ms,', d['iters'], 'iterations')"
  DIAGLIB_AMD_NO_PENDING=1 DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --roots 16 --steps 10 --warmup 3 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('davidson n=2e6 16 roots n_max=21, pending blocks off:', d['ms_per_step'], 'ms,', d['iters'], 'iterations')"
done > $OUT/davidson_cfg4_pending_ab.txt 2>&1 || true
for nn in 50000 250000; do DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --n $nn --steps 30 --warmup 3 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('shard of $nn rows:', d['ms_per_step'], 'ms per solve,', d['iters'], 'iterations,', d['host_syncs'] // d['steps'], 'host waits')"; done > $OUT/shard_floor.txt 2>&1 || true
# drop-in mode: time inside the caller's routine / waiting for downloads / waiting for small results against the chunk count
for c in 1 2 4 8; do
  echo "== DLA_OPT_STAGE_CHUNKS = $c (two solves; api lines = totals over both)" >> $OUT/host_mode_chunks.txt
  DIAGLIB_AMD_HOSTTIME=1 python3 tools/host_mode_timeline.py 2000000 $c 2>&1 | grep -E "solve 1|user callback|stage d2h \(wait\)|small result|alone" >> $OUT/host_mode_chunks.txt
done
python3 tools/lr_gen_bench.py --ab-run-ahead 2>/dev/null | grep "run-ahead" > $OUT/lr_gen_run_ahead_ab.txt
python3 tools/fuzz_run_ahead.py 16 7 > $OUT/fuzz_run_ahead.txt 2>&1 || true
mkdir -p tools/bin && hipcc --offload-arch=gfx950 -O2 -o tools/bin/upload_probe tools/upload_probe.hip && timeout -k 5 60 tools/bin/upload_probe > $OUT/upload_probe.txt 2>&1 || true
python3 tools/ritz_skew_probe.py > $OUT/ritz_skew_probe.txt 2>&1 || true
for r in 1 2 3; do
  DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('davidson n=2e6 8 roots, pending factors on :', d['ms_per_step'], 'ms,', d['iters'], 'iterations')"
  DIAGLIB_AMD_NO_PENDING=1 DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('davidson n=2e6 8 roots, pending factors off:', d['ms_per_step'], 'ms,', d['iters'], 'iterations')"
done > $OUT/davidson_pending_ab.txt 2>&1 || true
(python3 tools/lobpcg_pending_ab.py 2000000 8 2e-13; python3 tools/lobpcg_pending_ab.py 10000000 32 1e-12) 2>/dev/null | grep "pending" > $OUT/lobpcg_pending_ab.txt || true
hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/rr_probe.hip -Ldiaglib_amd/lib -ldiaglib_amd -Wl,-rpath,$PWD/diaglib_amd/lib -o tools/bin/rr_probe && timeout -k 5 120 tools/bin/rr_probe 20 > $OUT/rr_device_probe.txt 2>&1 || true
echo "rehearsal / lr / host done"
fi
if [[ $PART == *C* ]]; then
bash tools/profile_stalls.sh headline > /dev/null 2>&1 && cp gpurun_out/stall_counters_headline.txt $OUT/stall_counters_headline.txt
bash tools/profile_stalls.sh cfg5 --solver lobpcg --n 10000000 --roots 32 --tol 1e-12 > /dev/null 2>&1 && cp gpurun_out/stall_counters_cfg5.txt $OUT/stall_counters_cfg5shape.txt
rm -rf gpurun_out/stalls_headline gpurun_out/stalls_cfg5
echo "stall counters done"
python3 tools/floor_probe.py --n 10000000 --roots 32 --solver lobpcg --iters 30 --impl hip > $OUT/floor_probe_lobpcg_n1e7_32roots_hip.txt 2>/dev/null
python3 tools/floor_probe.py --n 2000000 --roots 8 --solver davidson --iters 16 --impl hip > $OUT/floor_probe_davidson_n2e6_8roots_hip.txt 2>/dev/null
python3 tools/floor_probe.py --n 1000000 --roots 32 --solver lobpcg --iters 40 --impl hip,oracle,reference > $OUT/floor_probe_lobpcg_n1e6_32roots.txt 2>/dev/null
rm -f $OUT/lr_gen.err
fi
if [[ $PART == *D* ]]; then
# the schedules of ortho_vs_x, interleaved in one call: tune knob 6 = 0 (shipped: pending blocks on the device too, first factor from the
# projected block's Gram matrix), 15 (the same with the first factor from U^T U), 14 (dla_expand_project mode 5 behaves like mode 4: the
# round's earlier state, tight bounds on what stays pending), 12 (five-sweep schedule) -- headline + random-guess leg, LOBPCG cfg 3, cfg 5 shape
for r in 1 2 3; do for t in 0 15 14 12; do
  DIAGLIB_BENCH_TUNE="6=$t" python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('davidson n=2e6 8 roots  knob6=$t:', d['ms_per_step'], 'ms,', d['config']['iters'], 'iterations,', d['roofline']['solve']['alg_GB'], 'GB, step', d['roofline']['step']['frac'], '| random-guess leg', d['config']['random_guess_leg']['ms'], 'ms,', d['config']['random_guess_leg']['iters'], 'iterations')"
done; done > $OUT/schedule_ab.txt 2>&1 || true
for r in 1 2; do for t in 0 15 12; do
  DIAGLIB_BENCH_TUNE="6=$t" python3 bench.py --solver lobpcg --steps 10 --warmup 3 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lobpcg n=2e6 8 roots    knob6=$t:', d['ms_per_step'], 'ms,', d['config']['iters'], 'iterations,', d['roofline']['solve']['alg_GB'], 'GB')"
done; done >> $OUT/schedule_ab.txt 2>&1 || true
for r in 1 2; do for t in 0 12; do
  DIAGLIB_BENCH_TUNE="6=$t" python3 bench.py --solver lobpcg --n 10000000 --roots 32 --tol 1e-12 --steps 3 --warmup 1 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lobpcg n=1e7 32 roots   knob6=$t:', d['ms_per_step'], 'ms,', d['config']['iters'], 'iterations,', d['roofline']['solve']['alg_GB'], 'GB')"
done; done >> $OUT/schedule_ab.txt 2>&1 || true
python3 tools/solve_jitter.py 30 > $OUT/solve_jitter.txt 2>/dev/null || true
python3 tools/solve_jitter.py 30 14 >> $OUT/solve_jitter.txt 2>/dev/null || true
DIAGLIB_AMD_CHAIN_DEBUG=1 DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-random-leg 2>&1 | grep "chain k=" | tail -8 | cut -c1-120 > $OUT/chain_schedules_headline.txt
python3 tools/determinism_probe.py 2000000 10 > $OUT/determinism.txt 2>/dev/null || true
echo "schedule A/B done"
fi
if [[ $PART == *A* ]]; then
python3 - <<PY
import json
a = json.load(open("$OUT/pmc_traffic_headline.json")); c = json.load(open("$OUT/pmc_traffic_cfg5shape.json"))
k1 = "davidson n=2000000 roots=8 n_max=13 max_dav=20 guess=unit"; k5 = "lobpcg n=10000000 roots=32 n_max=37 max_dav=20 guess=unit"
json.dump({k1: a[k1], k5: c[k5]}, open("$OUT/pmc_traffic.json", "w"), indent=1, sort_keys=True)
PY
rm -f $OUT/pmc_traffic_headline.json $OUT/pmc_traffic_cfg5shape.json
rm -rf $P/kt $P/mfma $P/fetch $P/write gpurun_out/profile_${TAG}c5
fi
ls $OUT
