#!/bin/bash
cd "$(dirname "$0")/../.." || exit 1
export MFVI_TUNE_CACHE=/tmp/tunes_graph_ab.json
for spec in "--k 1" "--config cfg1"; do
  for env in "MFVI_GRAPH_SIDE=1" "MFVI_GRAPH_SIDE=0" "MFVI_GRAPH_SIDE=0 MFVI_FWD_FORK=0"; do
    env $env python3 bench.py $spec --graph --no-cpu-baseline --no-gpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-16s graph %-36s %.4f ms/iteration' % ('$spec', '$env', r['ms_per_step']))"
  done
  MFVI_SIDE_STREAM=0 python3 bench.py $spec --no-cpu-baseline --no-gpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-16s eager, side stream off: %.4f ms/iteration' % ('$spec', r['ms_per_step']))"
done
