#!/bin/bash
# eager vs graph-replayed iteration: K = 1 at 256^2 (the reference's own loop shape), cfg1, cfg2
cd "$(dirname "$0")/../.." || exit 1
out=$1; : > $out
export MFVI_TUNE_CACHE=/tmp/tunes_graph_ab.json
for spec in "--k 1" "--config cfg1" ""; do
  for g in "" "--graph"; do
    python3 bench.py $spec $g --no-cpu-baseline --no-gpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print('%-16s %-8s %.4f ms/iteration  %.1f it/s' % ('$spec' or 'cfg2', '$g' or 'eager', r['ms_per_step'], r['elbo_iters_per_sec']))" >> $out
  done
done
cat $out
