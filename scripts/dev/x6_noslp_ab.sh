#!/bin/bash
# -fno-slp-vectorize for the forward / backward-weight bf16x6 kernels (packed fp32 beside a matrix stream): A/B on one box
cd "$(dirname "$0")/../.." || exit 1
echo "=== as built"; python3 scripts/dev/x6_layers.py 2>/dev/null | grep -E "autotuned"
for f in conv_x6 conv_bww_x6 conv_rp; do
  scripts/dev/build_variant.sh $f /tmp/lib_ns_$f.so -fno-slp-vectorize 2>/dev/null || exit 1
  echo "=== $f with -fno-slp-vectorize"; MFVI_LIB_PATH=/tmp/lib_ns_$f.so python3 scripts/dev/x6_layers.py 2>/dev/null | grep -E "autotuned"
done
