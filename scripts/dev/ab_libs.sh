#!/bin/bash
# whole-iteration A/B of several library builds on one box: ms per ELBO iteration (two rounds) + per-kernel table rows matching $PAT
PAT=${PAT:-"concat|sum of kernel"}
export MFVI_TUNE_CACHE=$PWD/gpurun_out/ab_tunes.json
python3 bench.py --no-cpu-baseline --steps 5 > /dev/null 2>&1
for rep in 1 2; do
for lib in "$@"; do
  MFVI_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --steps 40 --warmup 5 --profile-all 2> /tmp/ab_err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-60s ms_per_step %.4f' % ('$lib'.split('/')[-1], d['ms_per_step']))"
  [ $rep = 1 ] && grep -E "$PAT" /tmp/ab_err.txt
done
done
