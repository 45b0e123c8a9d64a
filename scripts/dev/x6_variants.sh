#!/bin/bash
# times the bf16x6 backward-weight kernel for library variants built by build_variant.sh (X6_VARIANTS = suffixes of mfvi-dip-mia_amd/libvar_*.so)
for v in "" ${X6_VARIANTS:-NOPROD NOMFMA}; do
  lib=mfvi-dip-mia_amd/libmfvi_hip.so; [ -n "$v" ] && lib=mfvi-dip-mia_amd/libvar_$v.so
  echo "== variant ${v:-full} =="
  MFVI_LIB_PATH=$PWD/$lib X6_AUTOTUNE=0 timeout -k 10 300 python scripts/dev/x6_layers.py 2>&1 | grep "bf16x6 cof=[12] tb=1"
done
