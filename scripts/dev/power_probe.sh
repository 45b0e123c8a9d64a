#!/bin/bash
# samples clocks / power of the GPU while the bench loop runs (is the iteration power- or clock-limited?)
python3 bench.py --no-cpu-baseline --steps 3000 --warmup 5 > /tmp/pp.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|Power|GPU use|busy" | tr '\n' ';' ; echo
  sleep 1
done
wait $BP
python3 -c "import json; d=json.load(open('/tmp/pp.json')); print('ms_per_step', d['ms_per_step'])"
echo "idle:"; sleep 3; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ';'; echo
