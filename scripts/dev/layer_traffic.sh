#!/bin/bash
# HBM bytes per launch of ONE layer's kernels (alone on the stream): FETCH_SIZE / WRITE_SIZE in separate PMC passes + a plain trace,
# joined by scripts/hbm_traffic.py.  usage: layer_traffic.sh out.txt cin cout k s H W
export TMPDIR=/tmp
R=$(pwd); out=$R/$1; shift
export MFVI_TUNE_CACHE=/tmp/lt_tunes.json; rm -f $MFVI_TUNE_CACHE
python3 scripts/bench_layer.py "$@" 16 4 > /dev/null 2>&1
cd /tmp
rm -rf /tmp/lt_f /tmp/lt_w /tmp/lt_t
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/lt_f -- python3 $R/scripts/bench_layer.py "$@" 16 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/lt_w -- python3 $R/scripts/bench_layer.py "$@" 16 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d /tmp/lt_t -- python3 $R/scripts/bench_layer.py "$@" 16 4 > /dev/null 2>&1
echo "== layer $@ (K=16), every kernel alone on the stream" >> $out
python3 $R/scripts/hbm_traffic.py /tmp/lt_f /tmp/lt_w /tmp/lt_t 2>&1 | grep -E "^#|conv_" >> $out
