#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> <out dir>   (on the GPU box, in its scratch copy of the tree:
# the other build is swapped in for the runs marked "other" and the current one is put back)
set -e
other=$1; out=$2
cur=fa-vae_amd/favae_hip/libfavae_hip.so
mkdir -p "$out"
cp "$cur" "$out/cur.so"
run() { timeout -k 10 120 python -u tools/bench_grouped.py 16384 bf16 1,8 > "$out/$1.log" 2>&1; echo "$1: $(grep G= "$out/$1.log" | tr '\n' ' ')"; }
for r in 1 2; do
  cp "$out/cur.so" "$cur"; run "current_$r"
  cp "$other" "$cur"; run "other_$r"
done
cp "$out/cur.so" "$cur"; rm -f "$out/cur.so"
