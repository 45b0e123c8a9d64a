#!/bin/bash
# HBM bytes per launch of the row-phase backward-data kernel of one layer for several forced tilings (MFVI_TUNE_RP=mf,r,T,rem; no autotune):
# usage: rp_traffic_tilings.sh out.txt cin cout H W "mf,r,T,rem" ...
export TMPDIR=/tmp
R=$(pwd); out=$R/$1; cin=$2; cout=$3; H=$4; W=$5; shift 5
for t in "$@"; do
  export MFVI_TUNE_RP=$t AUTOTUNE=0
  cd /tmp; rm -rf /tmp/rt_f /tmp/rt_w /tmp/rt_t
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/rt_f -- python3 $R/scripts/bench_layer.py $cin $cout 3 1 $H $W 16 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/rt_w -- python3 $R/scripts/bench_layer.py $cin $cout 3 1 $H $W 16 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv -d /tmp/rt_t -- python3 $R/scripts/bench_layer.py $cin $cout 3 1 $H $W 16 4 > /dev/null 2>&1
  echo "== MFVI_TUNE_RP=$t" >> $out
  python3 $R/scripts/hbm_traffic.py /tmp/rt_f /tmp/rt_w /tmp/rt_t 2>&1 | grep -E "conv_rp" >> $out
  cd $R
done
