#!/bin/bash
# Recompile ONE translation unit of libmfvi_hip.so and relink (seconds instead of the 5-minute full build).  usage: rebuild_one.sh conv_bww_x6 [extra hipcc flags]
set -e
cd "$(dirname "$0")/../../mfvi-dip-mia_amd"
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c csrc/$f.hip -o build/$f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmfvi_hip.so build/conv_fwd.o build/conv_bwd_data.o build/conv_bwd_weight.o build/conv_mfma.o build/conv_rp.o build/conv_x6.o build/conv_small.o build/conv_bww_mfma.o build/conv_bww_x6.o build/elementwise.o build/losses.o build/radon.o build/plan.o
echo "relinked libmfvi_hip.so"
