#!/bin/bash
# the backward pass of one concat layer alone on the stream vs with the side stream (backward-weight beside backward-data), per library
for lib in "$@"; do
  for shape in "36 16 3 1 256 256" "68 32 3 1 128 128"; do
    for alone in 1 0; do
      echo "== $(basename $lib) $shape ALONE=$alone"
      MFVI_LIB_PATH=$lib ALONE=$alone python3 scripts/bench_layer.py $shape 16 20 2>/dev/null | grep -E "tunes|bwd_weight|bwd_data"
    done
  done
done
