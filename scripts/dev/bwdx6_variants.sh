#!/bin/bash
# Timing-only variants of the bf16x6 backward-data kernel (results wrong by design; each is its own .so next to the product library):
# where does a strip's time go?  usage (GPU box): scripts/dev/bwdx6_variants.sh out.txt     (variants are built on the box: hipcc is there)
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
for v in NONE NOSTAGE NOFOLDLD NOFOLDST "NOFOLDLD -DX6B_DBG_NOFOLDST" NOMFMA NOWCOPY "NOSTAGE -DX6B_DBG_NOFOLDLD -DX6B_DBG_NOFOLDST -DX6B_DBG_NOWCOPY"; do
  name=$(echo $v | tr -d ' ' | sed 's/-DX6B_DBG_/_/g')
  scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_$name.so -DX6B_DBG_$v || exit 1
  echo "=== variant $name" >> $out
  MFVI_LIB_PATH=/tmp/lib_$name.so BWDX6_ONLY=1 python3 scripts/dev/bwdx6_layers.py 2>/dev/null | grep -E "bf16x6|autotuned" >> $out
done
