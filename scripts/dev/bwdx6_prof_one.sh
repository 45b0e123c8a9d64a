#!/bin/bash
# in-kernel phase profile of ONE timing-only variant: bwdx6_prof_one.sh "<flags>"
cd "$(dirname "$0")/../.." || exit 1
scripts/dev/build_variant.sh conv_bwd_x6 /tmp/lib_pv.so -DX6B_PROF $1 2>/dev/null || exit 1
echo "=== $1"; MFVI_LIB_PATH=/tmp/lib_pv.so python3 scripts/dev/bwdx6_prof.py 2>/dev/null | grep -E "us|s\.(fold|stage|wait|weight) |m\.(rows|wait|weight)"
