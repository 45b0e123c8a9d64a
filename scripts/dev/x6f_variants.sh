#!/bin/bash
# times the bf16x6 forward kernel for the timing-only variants built by build_variant.sh (results of those builds are wrong by design)
for v in "" ${X6_VARIANTS:-NOPROD NOMFMA NOBOTH}; do
  lib=mfvi-dip-mia_amd/libmfvi_hip.so; [ -n "$v" ] && lib=mfvi-dip-mia_amd/libvar_$v.so
  echo "== variant ${v:-full} =="
  MFVI_LIB_PATH=$PWD/$lib X6_AUTOTUNE=0 timeout -k 10 300 python scripts/dev/x6_layers.py 2>&1 | grep -E "fwd +bf16x6 mf=(1 sr=8 T=4|2 sr=8 T=1)"
done
