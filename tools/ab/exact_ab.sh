# interleaved A/B of schedule knobs on the headline: bash tools/exact_ab.sh "0 14 15" [rounds]
KN=${1:-"0 14"}; R=${2:-3}
for r in $(seq $R); do for t in $KN; do
  DIAGLIB_BENCH_TUNE="6=$t" timeout -k 10 200 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('knob6=$t:', d['ms_per_step'], 'ms,', d['config']['iters'], 'it,', d['roofline']['solve']['alg_GB'], 'GB | random leg', d['config']['random_guess_leg']['ms'], 'ms,', d['config']['random_guess_leg']['iters'], 'it')"
done; done
