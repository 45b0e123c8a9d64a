#!/bin/bash
# compiler-flag A/B of the bf16x6 backward-data kernel (one translation unit rebuilt per variant)
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
i=0
for v in "-DX6B_NOP=0" "-fno-slp-vectorize" "-fno-slp-vectorize -mllvm -amdgpu-enable-packed-math=0"; do
  i=$((i+1))
  scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_f$i.so $v 2>/dev/null || { echo "build failed: $v" >> $out; continue; }
  echo "=== flags: $v" >> $out
  MFVI_LIB_PATH=/tmp/lib_f$i.so BWDX6_ONLY=1 python3 scripts/dev/bwdx6_layers.py 2>/dev/null | grep -E "bf16x6" >> $out
done
