#!/bin/bash
# in-kernel phase profile of the strip-resident kernel, optionally of a timing-only variant: bwdx6s_prof.sh out.txt ["<extra flags>"]
cd "$(dirname "$0")/../.." || exit 1
scripts/dev/build_variant.sh conv_bwd_x6s /tmp/lib_xs.so -DX6S_PROF $2 || exit 1
MFVI_LIB_PATH=/tmp/lib_xs.so python3 scripts/dev/bwdx6s_prof.py 36 256 8 36 512 32 2>/dev/null > $1
