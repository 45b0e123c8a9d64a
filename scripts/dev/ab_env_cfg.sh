#!/bin/bash
# like ab_env.sh for another bench config: scripts/dev/ab_env_cfg.sh <cfg> "VAR=a" "VAR=b" ...
CFG=$1; shift
export MFVI_TUNE_CACHE=$PWD/gpurun_out/abc_tunes_$CFG.json
python3 bench.py --config $CFG --no-cpu-baseline --steps 5 > /dev/null 2>&1
for rep in 1 2; do
for setting in "$@"; do
  ( [ "$setting" != "-" ] && export $setting; python3 bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 5 2> /dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-6s %-44s ms_per_step %.4f' % ('$CFG', '$setting', d['ms_per_step']))" )
done
done
