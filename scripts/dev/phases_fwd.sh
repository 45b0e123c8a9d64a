#!/bin/bash
# per-phase cycle counts of one block of the forward / backward-data kernels (instrumented copy of the library, see prof_patch_conv_mfma.py)
export MFVI_LIB_PATH=$PWD/mfvi-dip-mia_amd/build/ab/libmfvi_prof.so MFVI_PROF=1 MFVI_SIDE_STREAM=0
OUT=${1:-gpurun_out/phases_fwd.txt}; : > $OUT
for spec in "36 16 3 1 256 256|1,16,4" "36 16 3 1 256 256|3,8,8" "68 32 3 1 128 128|1,16,4" "68 32 3 1 128 128|1,8,8"; do
  shape=${spec%%|*}; tune=${spec##*|}
  echo "== $shape MFVI_TUNE=$tune" >> $OUT
  MFVI_TUNE=$tune timeout -k 10 120 python3 scripts/bench_layer.py $shape 16 3 >> $OUT 2>&1 || exit 1
done
