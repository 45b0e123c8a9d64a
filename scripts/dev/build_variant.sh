#!/bin/bash
# Build a variant of the library that differs in ONE translation unit: build_variant.sh conv_bww_x6 out.so [-DFLAG ...]   (the other objects come from mfvi-dip-mia_amd/build)
set -e
f=$1; out=$2; shift; shift
d=mfvi-dip-mia_amd
ff=$(cd $d && python3 -c "import _build; print(' '.join(_build.FILE_FLAGS.get('$f.hip', [])))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $ff "$@" -c $d/csrc/$f.hip -o /tmp/${f}_variant_$$.o
objs=$(ls $d/build/*.o | grep -v "/$f.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs /tmp/${f}_variant_$$.o
rm -f /tmp/${f}_variant_$$.o
