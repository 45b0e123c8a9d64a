#!/bin/bash
# backward-weight of the three concat layers: autotuned choice vs the fragment-split variant at several block-count targets
for shape in "36 16 3 1 256 256" "68 32 3 1 128 128" "132 64 3 1 64 64" "128 128 3 1 32 32"; do
  echo "== $shape"
  python3 scripts/bench_layer.py $shape 16 20 2>/dev/null | grep -E "tunes|bwd_weight"
  for tb in 1 2 3 4; do echo -n "split tgt=$tb: "; MFVI_TUNE_W=2,10,$tb python3 scripts/bench_layer.py $shape 16 20 2>/dev/null | grep bwd_weight; done
done
