#!/bin/bash
# Timing-only variants of the strip-resident kernel (results wrong by design; each is its own .so): is a strip's time memory traffic?
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
for v in NONE NOY NOX NOFOLDST "NOY -DX6S_DBG_NOX -DX6S_DBG_NOFOLDST"; do
  name=$(echo $v | tr -d ' ' | sed 's/-DX6S_DBG_/_/g')
  scripts/dev/build_variant.sh conv_bwd_x6s /tmp/lib_$name.so -DX6S_DBG_$v || exit 1
  echo "=== variant $name" >> $out
  MFVI_LIB_PATH=/tmp/lib_$name.so BWDX6_ONLY=1 python3 scripts/dev/bwdx6_layers.py 36 16 256 2>/dev/null | grep -E "strip-resident T= 8" >> $out
done
