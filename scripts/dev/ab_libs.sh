#!/bin/bash
# whole-iteration A/B of several library builds on one box: ms per ELBO iteration (two rounds) + per-kernel table rows matching $PAT.
# Every build autotunes for itself (its own cache file), so a build with new tiling candidates is measured with them.
PAT=${PAT:-"concat|sum of kernel"}
CFG=${CFG:-cfg2}
for lib in "$@"; do
  MFVI_TUNE_CACHE=$PWD/gpurun_out/ab_tunes_$(basename $lib).json MFVI_LIB_PATH=$lib python3 bench.py --config $CFG --no-cpu-baseline --steps 5 > /dev/null 2>&1
done
for rep in 1 2; do
for lib in "$@"; do
  MFVI_TUNE_CACHE=$PWD/gpurun_out/ab_tunes_$(basename $lib).json MFVI_LIB_PATH=$lib python3 bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 5 --profile-all 2> /tmp/ab_err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-8s %-40s ms_per_step %.4f  roofline frac %.3f alone %.3f' % ('$CFG', '$lib'.split('/')[-1], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('alone', {}).get('frac', 0) if isinstance(d['roofline'].get('alone'), dict) else 0))"
  [ $rep = 1 ] && grep -E "$PAT" /tmp/ab_err.txt
done
done
