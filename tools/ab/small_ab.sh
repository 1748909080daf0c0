for n in 50000 250000; do for r in 1 2; do for t in 0 15 14; do
DIAGLIB_BENCH_TUNE="6=$t" DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --n $n --steps 30 --warmup 3 --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('n=$n knob6=$t:', d['ms_per_step'], 'ms', d['iters'], 'it', d['host_syncs'])"
done; done; done
