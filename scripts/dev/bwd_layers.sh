#!/bin/bash
# backward-data of the three concat layers, alone, per library
for lib in "$@"; do
  for shape in "36 16 3 1 256 256" "68 32 3 1 128 128" "132 64 3 1 64 64"; do
    echo -n "$(basename $lib) "; MFVI_LIB_PATH=$lib python3 scripts/bench_layer.py $shape 16 20 2>/dev/null | grep -E "bwd_data"
  done
done
