#!/bin/bash
# A/B of one FAVAE_FUSE_MASK bit on the same box: tools/ab_fuse.sh <bit (hex)> <out dir> [n]
# (bench_grouped.py's bf16 rows, alternating default / bit set, three rounds)
set -e
bit=$1; out=$2; n=${3:-16384}
mkdir -p "$out"
for r in 1 2 3; do
  FAVAE_FUSE_MASK=0x17f timeout -k 10 120 python -u tools/bench_grouped.py "$n" bf16 1,8 > "$out/a$r.log" 2>&1
  FAVAE_FUSE_MASK=$(printf '0x%x' $((0x17f | bit))) timeout -k 10 120 python -u tools/bench_grouped.py "$n" bf16 1,8 > "$out/b$r.log" 2>&1
done
for r in 1 2 3; do echo "default:"; grep G= "$out/a$r.log"; echo "bit $bit set:"; grep G= "$out/b$r.log"; done
