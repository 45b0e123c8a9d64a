#!/bin/bash
# per-pass times of the small-map layers of the den net (K = 16), alone on the stream
for shape in "128 128 3 1 16 16" "128 128 3 1 8 8" "128 128 3 2 32 32" "128 128 3 2 16 16" "132 128 3 1 16 16" "132 128 3 1 32 32" "128 128 3 1 32 32" "128 128 1 1 16 16" "128 4 1 1 16 16" "64 128 3 2 64 64"; do
  python3 scripts/bench_layer.py $shape 16 30 2>/dev/null
done
