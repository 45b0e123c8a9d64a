#!/bin/bash
# who folds: the matrix waves from their accumulators (MFOLD = 1) or the staging waves through LDS (0); default: 1 for the 16-output-channel form only
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
for v in "-DX6B_NOP=0" "-DX6B_MFOLD=1" "-DX6B_MFOLD=0"; do
  scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_mf.so $v 2>/dev/null || exit 1
  echo "=== $v" >> $out
  MFVI_LIB_PATH=/tmp/lib_mf.so BWDX6_ONLY=1 python3 scripts/dev/bwdx6_layers.py 2>/dev/null | grep -E "bf16x6" >> $out
done
