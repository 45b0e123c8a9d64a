#!/bin/bash
# every BASELINE config through bench.py (one JSON line each) -> gpurun_out/r2_bench_<cfg>.json, per-kernel tables -> .err
for c in cfg2 cfg1 cfg3 cfg4 cfg5 inp; do
  python3 bench.py --config $c --profile-all > gpurun_out/r2_bench_$c.json 2> gpurun_out/r2_bench_$c.err || echo "FAILED $c"
  python3 -c "
import json; d=json.load(open('gpurun_out/r2_bench_$c.json')); r=d['roofline']
print('$c', 'ms/step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], '|', r['kernel'], 'frac %.3f alone %.3f' % (r['frac'], r['alone']['frac']), '| cpu', (d.get('cpu_baseline') or {}).get('value'))"
done
