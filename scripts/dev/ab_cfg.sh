#!/bin/bash
# A/B of library builds on another bench config: scripts/dev/ab_cfg.sh <cfg> lib1.so lib2.so ...
CFG=$1; shift
export MFVI_TUNE_CACHE=$PWD/gpurun_out/ab_tunes_$CFG.json
python3 bench.py --config $CFG --no-cpu-baseline --steps 5 > /dev/null 2>&1
for rep in 1 2; do
for lib in "$@"; do
  MFVI_LIB_PATH=$lib python3 bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 5 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-40s ms_per_step %.4f' % ('$CFG', '$lib'.split('/')[-1], d['ms_per_step']))"
done
done
