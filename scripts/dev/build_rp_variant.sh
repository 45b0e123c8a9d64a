#!/bin/bash
# Build a variant of the library that differs only in conv_rp.hip: build_rp_variant.sh out.so [-DFLAG ...]   (the other objects come from mfvi-dip-mia_amd/build)
set -e
out=$1; shift
d=mfvi-dip-mia_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c $d/csrc/conv_rp.hip -o /tmp/conv_rp_variant_$$.o
objs=$(ls $d/build/*.o | grep -v conv_rp.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs /tmp/conv_rp_variant_$$.o
