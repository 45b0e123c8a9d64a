#!/bin/bash
# Layer micro-benchmarks of the three dominant 3x3 layers with their autotuned tilings (fwd / bwd-data forced by MFVI_TUNE).
set -e
for spec in "36 16 1,8,8 1,8,4" "68 32 1,8,4 1,8,4" "132 64 2,136,1 1,144,1"; do
  set -- $spec
  hw=$((256 * 16 / $2)); [ $2 = 64 ] && hw=64; [ $2 = 32 ] && hw=128; [ $2 = 16 ] && hw=256
  MFVI_TUNE=$3 python3 scripts/bench_layer.py $1 $2 3 1 $hw $hw 16 10 | grep -E "fwd" || true
  MFVI_TUNE=$4 python3 scripts/bench_layer.py $1 $2 3 1 $hw $hw 16 10 | grep -E "bwd_data|bwd_weight" || true
done
