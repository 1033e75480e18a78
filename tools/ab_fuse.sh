# A/B of tuning knobs on one box (developer tool): prints ms/step of the bf16 training epoch per setting
run() { echo "$1"; env $1 timeout -k 10 100 python tools/prof_step.py bf16 16384 2 2>/dev/null | grep epoch | tail -1 || exit 1; }
for s in "$@"; do run "$s" || exit 1; done
