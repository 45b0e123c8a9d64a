export TMPDIR=/tmp
R=$(pwd)
export MFVI_TUNE_CACHE=$R/gpurun_out/tunes_k1.json
python3 bench.py --k 1 --no-cpu-baseline --steps 20 > /dev/null 2>&1
cd /tmp && rm -rf /tmp/p_k1t && rocprofv3 --kernel-trace --output-format csv -d /tmp/p_k1t -- python3 $R/bench.py --k 1 --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python3 $R/scripts/timeline.py $(find /tmp/p_k1t -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/timeline_k1.txt
