#!/bin/bash
# in-kernel phase profile of timing-only variants (results wrong by design)
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
for v in NONE NOSTAGE NOXREAD "NOSTAGE -DX6B_DBG_NOXREAD" "NOSTAGE -DX6B_DBG_NOFOLDLD -DX6B_DBG_NOFOLDST -DX6B_DBG_NOWCOPY" "NOSTAGE -DX6B_DBG_NOFOLDLD -DX6B_DBG_NOFOLDST -DX6B_DBG_NOWCOPY -DX6B_DBG_NOXREAD" "NOMFMA"; do
  name=$(echo $v | tr -d ' ' | sed 's/-DX6B_DBG_/_/g')
  scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_pv.so -DX6B_PROF -DX6B_DBG_$v 2>/dev/null || exit 1
  echo "=== variant $name" >> $out
  MFVI_LIB_PATH=/tmp/lib_pv.so python3 scripts/dev/bwdx6_prof.py 2>/dev/null | grep -E "T=|m\.rows|m\.fold|m\.weight|m\.wait|m\.total|m\.prol" >> $out
done
