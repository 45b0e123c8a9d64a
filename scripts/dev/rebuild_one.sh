#!/bin/bash
# Recompile ONE translation unit of libmfvi_hip.so with the product flags and relink (seconds instead of the 5-minute full build).
# usage: rebuild_one.sh conv_bww_x6
# Extra hipcc flags are refused here: a timing-only or debug variant (-DX6_DBG_NOPROD, -DRP_DBG_NOLOAD ... "results wrong by design") must
# never stay behind as the product library — _build.py and bench.py identify a build by its SOURCES only.  Variants go through
# build_variant.sh (own output file) + MFVI_LIB_PATH.
set -e
cd "$(dirname "$0")/../../mfvi-dip-mia_amd"
f=$1; shift
if [ $# -gt 0 ]; then
    echo "rebuild_one.sh: extra flags ($*) are not allowed for libmfvi_hip.so; use scripts/dev/build_variant.sh $f <out.so> $*" >&2
    exit 2
fi
objs=$(python3 -c "import _build; print(' '.join('build/' + s.replace('.hip', '.o') for s in _build.SOURCES))")
flags=$(python3 -c "import _build; print(' '.join(_build.FLAGS + _build.FILE_FLAGS.get('$f.hip', [])))")
/opt/rocm/bin/hipcc $flags -c csrc/$f.hip -o build/$f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libmfvi_hip.so $objs
echo "relinked libmfvi_hip.so"
