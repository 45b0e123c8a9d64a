#!/bin/bash
# Is the forward 3x3 kernel bound by the latency of its staged global loads?  BN=0 feeds the layer from the shared net input
# (sample stride 0: all 16 samples read the same 9 MB, L2 / Infinity-Cache hits), BN=1 from a per-sample tensor (151 MB from HBM).
for bn in 1 0; do
  for tune in 1,8,8 1,8,4 1,8,2 1,8,1 1,16,4 1,16,1; do
    echo -n "BN=$bn tune=$tune  "; BN=$bn MFVI_TUNE=$tune python3 scripts/bench_layer.py 36 16 3 1 256 256 16 10 2>/dev/null | grep -E "fwd" | sed 's/.*fwd/fwd/'
  done
done
