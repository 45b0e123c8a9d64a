#!/bin/bash
# whole-iteration A/B of environment switches of ONE library build, all in one process chain on one box (boxes differ by ~3%):
#   scripts/dev/ab_env.sh "VAR=a" "VAR=b" ...     (each argument: space-separated VAR=value settings; "-" = none)
export MFVI_TUNE_CACHE=$PWD/gpurun_out/ab_tunes.json
python3 bench.py $AB_ARGS --no-cpu-baseline --no-gpu-baseline --steps 5 > /dev/null 2>&1      # fills the autotune cache
for rep in 1 2 3; do
for setting in "$@"; do
  ( [ "$setting" != "-" ] && export $setting; python3 bench.py $AB_ARGS --no-cpu-baseline --no-gpu-baseline --steps ${AB_STEPS:-200} --warmup 5 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s ms_per_step %.4f' % ('$setting', d['ms_per_step']))" )
done
done
