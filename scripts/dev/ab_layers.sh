#!/bin/bash
# A/B of two builds of the library on the three dominant 3x3 layers (autotuned tilings of the bench) and on the whole iteration.
# usage: ab_layers.sh libA.so libB.so
for lib in "$@"; do
  echo "=== $lib"
  export MFVI_LIB_PATH=$lib
  for spec in "36 16 1,8,8 1,8,4 256" "68 32 1,8,4 1,8,4 128" "132 64 2,136,1 1,144,1 64"; do
    set -- $spec
    MFVI_TUNE=$3 python3 scripts/bench_layer.py $1 $2 3 1 $5 $5 16 10 | grep -E "fwd" || true
    MFVI_TUNE=$4 python3 scripts/bench_layer.py $1 $2 3 1 $5 $5 16 10 | grep -E "bwd_data|bwd_weight" || true
  done
  python3 bench.py --no-cpu-baseline --steps 30 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'], 'fwd_only', d['fwd_only_mc_passes_per_sec'], d['roofline']['alone'])"
done
