#!/bin/bash
# same-box sweep of one environment knob of the executor: tools/sweep_env.sh NAME "v1 v2 ..." <out dir> [groups]
# (bench_grouped.py bf16 rows; the unset default first and last)
set -e
name=$1; vals=$2; out=$3; g=${4:-1}
mkdir -p "$out"
run() { timeout -k 10 120 python -u tools/bench_grouped.py 16384 bf16 "$g" > "$out/$1.log" 2>&1; echo "$1: $(grep G= "$out/$1.log" | tr '\n' ' ')"; }
run default_a
for v in $vals; do export $name=$v; run "${name}_$v"; unset $name; done
run default_b
