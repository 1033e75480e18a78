# A/B of knobs on the one-rank rehearsal of the data-parallel path (developer tool): samples/s of bench.py at N = 32768
run() { echo "$1"; env $1 FAVAE_FORCE_DP=1 timeout -k 10 200 python -u bench.py --samples 32768 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  %.0f samples/s  %.1f ms per iteration' % (d['value'], d['ms_per_step']))" || exit 1; }
for s in "$@"; do run "$s" || exit 1; done
