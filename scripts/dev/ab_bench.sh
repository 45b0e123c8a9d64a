#!/bin/bash
# whole-iteration A/B of several library builds: ms per ELBO iteration + the per-op table of the dominant backward-data kernels
for lib in "$@"; do
  echo "=== $lib"
  MFVI_LIB_PATH=$lib python3 bench.py --no-cpu-baseline --steps 30 --profile-all 2> /tmp/ab_err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],4), 'fwd_only', round(d['fwd_only_mc_passes_per_sec']))"
  grep -E "bwd_data 3x3 |bwd_data 1x1|op 28 bwd_data|op 25 bwd_data|op 22 bwd_data|op 19 bwd_data|op 29 bwd_data|sum of kernel" /tmp/ab_err.txt
done
