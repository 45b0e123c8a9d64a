#!/bin/bash
# per-phase cycle counts of one block of the forward / backward-data kernels (instrumented copy of the library, see prof_patch_conv_mfma.py)
# usage: phases_fwd.sh out.txt "cin cout ks stride H W|mf,th,T" ...
export MFVI_LIB_PATH=$PWD/mfvi-dip-mia_amd/build/ab/libmfvi_prof.so MFVI_PROF=1 MFVI_SIDE_STREAM=0
OUT=$1; shift; : > $OUT
for spec in "$@"; do
  shape=${spec%%|*}; tune=${spec##*|}
  echo "== $shape MFVI_TUNE=$tune" >> $OUT
  MFVI_TUNE=$tune timeout -k 10 120 python3 scripts/bench_layer.py $shape 16 3 >> $OUT 2>&1 || exit 1
done
