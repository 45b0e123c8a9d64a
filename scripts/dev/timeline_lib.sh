#!/bin/bash
# two-stream timeline of one traced iteration for a given library build: scripts/dev/timeline_lib.sh <lib.so> <out.txt>
export TMPDIR=/tmp
R=$(pwd)
export MFVI_LIB_PATH=$1
export MFVI_TUNE_CACHE=$R/gpurun_out/tl_tunes_$(basename $1).json
python3 bench.py --steps 5 --no-cpu-baseline > /dev/null 2>&1
cd /tmp && rm -rf /tmp/p_tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_tl -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cd $R && python3 scripts/timeline.py $(find /tmp/p_tl -name "*kernel_trace.csv" | head -1) > $2 2>&1
