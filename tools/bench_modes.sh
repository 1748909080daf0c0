for i in 1 2; do
python3 bench.py 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default:', d['ms_per_step'], d['steps'], d['warmup'])"
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no cpu baseline:', d['ms_per_step'], d['steps'], d['warmup'])"
python3 bench.py --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no cpu baseline, no random leg:', d['ms_per_step'], d['steps'], d['warmup'])"
DIAGLIB_BENCH_NOPROFILE=1 python3 bench.py --no-cpu-baseline --no-random-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no events:', d['ms_per_step'], d['steps'], d['warmup'])"
python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('10/3:', d['ms_per_step'], d['steps'], d['warmup'])"
done
