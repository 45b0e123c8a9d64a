#!/bin/bash
# Round-2 profile set (run on the GPU box; results under gpurun_out/prof_r02/, the summaries are then copied into profiles/):
#   1. autotune cache + bench JSON line of the default workload (cfg2)
#   2. rocprofv3 --kernel-trace --stats of the same command (autotune cache exported: no search launches in the profiled process)
#   3. HBM traffic: two PMC passes (FETCH_SIZE, WRITE_SIZE; counters + kernel trace only, side stream off) joined with launch durations
#   4. SQ counters of the 36->16 @256^2 layer's kernels (three PMC passes on scripts/bench_layer.py)
#   5. kernel stats of the other BASELINE configs
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_r02; rm -rf $OUT; mkdir -p $OUT
R=$(pwd)
export MFVI_TUNE_CACHE=$R/$OUT/tunes_cfg2.json      # absolute: the profiled commands run from /tmp
python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $R/$OUT/kernel_stats_cfg2.csv
cp $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) $R/$OUT/kernel_trace_cfg2.csv
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 $R/scripts/hbm_traffic.py /tmp/p_fetch /tmp/p_write /tmp/p_trace > $R/$OUT/pmc_hbm_traffic.txt 2>&1
cd $R
unset MFVI_TUNE_CACHE
rm -f $OUT/pmc_sq_up9.txt; scripts/dev/pmc_sq.sh $OUT/pmc_sq_up9.txt 36 16 3 1 256 256
for c in cfg1 cfg3 cfg4 cfg5 inp; do
  export MFVI_TUNE_CACHE=$R/$OUT/tunes_$c.json
  python3 bench.py --config $c > $OUT/bench_$c.json 2> /dev/null
  (cd /tmp && rm -rf /tmp/p_$c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -- python3 $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
  cp $(find /tmp/p_$c -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$c.csv
  echo "done $c"
done
ls -la $OUT
