#!/bin/bash
# wave priorities of the bf16x6 backward-data kernel (staging / matrix): A/B on one box
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
for v in "0 2" "1 0" "0 0" "0 3" "2 0"; do
  set -- $v
  scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_p$1$2.so -DX6B_PRIO_S=$1 -DX6B_PRIO_M=$2 || exit 1
  echo "=== staging prio $1, matrix prio $2" >> $out
  MFVI_LIB_PATH=/tmp/lib_p$1$2.so BWDX6_ONLY=1 python3 scripts/dev/bwdx6_layers.py 2>/dev/null | grep -E "bf16x6" >> $out
done
