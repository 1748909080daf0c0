for r in 1 2 3; do for t in 0 14; do
  DIAGLIB_BENCH_TUNE="6=$t" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knob6=$t:', d['ms_per_step'], 'ms,', d['config']['iters'], 'it,', d['roofline']['solve']['alg_GB'], 'GB, resid', d['config'].get('residual'), '| random leg', d['config']['random_guess_leg']['ms'], 'ms,', d['config']['random_guess_leg']['iters'], 'it')"
done; done
