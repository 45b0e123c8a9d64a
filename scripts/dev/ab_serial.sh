#!/bin/bash
# two library builds, side stream on / off, one box
for lib in "$@"; do
  export MFVI_LIB_PATH=$lib MFVI_TUNE_CACHE=$PWD/gpurun_out/abs_$(basename $lib).json
  python3 bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
  for rep in 1 2; do for ss in 1 0; do
    MFVI_SIDE_STREAM=$ss python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-16s side_stream=$ss ms_per_step %.4f' % ('$(basename $lib)', d['ms_per_step']))"
  done; done
done
