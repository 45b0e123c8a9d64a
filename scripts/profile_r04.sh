#!/bin/bash
# Round-4 profile set (run on the GPU box; results under gpurun_out/prof_r04/, the summaries are then copied into profiles/):
#   1. autotune cache + bench JSON line of the default workload (cfg2); K = 1 engine leg; drop-in legs (torch.optim.AdamW over 254 Parameters / one flat one)
#   2. rocprofv3 --kernel-trace --stats of the same command (autotune cache exported: no search launches in the profiled process) + timeline
#   3. HBM traffic: two PMC passes (FETCH_SIZE, WRITE_SIZE; counters + kernel trace only, side stream off) joined with launch durations
#   4. HBM traffic and SQ counters of the three dominant 3x3 layers, one at a time (every kernel alone on the stream)
#   5. bench line + kernel stats of the other BASELINE configs
#   6. the bf16x6 kernels over their tilings (x6_layers.py, bwdx6_layers.py) and the SQ counters of a layer they serve in two passes
#   6b. the strip-resident backward-data kernel (conv_bwd_x6s.hip): phase profile, timing-only variants
#   7. the graph-replayed iteration beside the eager one (graph_bench.sh); every bench line carries the unfused PyTorch-ROCm leg of its config
export TMPDIR=/tmp
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/prof_r04; rm -rf $OUT; mkdir -p $OUT
R=$(pwd)
export MFVI_TUNE_CACHE=$R/$OUT/tunes_cfg2.json      # absolute: the profiled commands run from /tmp
python3 bench.py > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python3 bench.py --profile-all --no-cpu-baseline --no-gpu-baseline --steps 20 2> $OUT/kernel_table_cfg2.txt > /dev/null
cd /tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write /tmp/p_trace      # (a reused box keeps /tmp: the copies below take the first match)
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_stats -name "*kernel_stats.csv" | head -1) $R/$OUT/kernel_stats_cfg2.csv
python3 $R/scripts/timeline.py $(find /tmp/p_stats -name "*kernel_trace.csv" | head -1) > $R/$OUT/timeline.txt
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1
MFVI_SIDE_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1
python3 $R/scripts/hbm_traffic.py /tmp/p_fetch /tmp/p_write /tmp/p_trace > $R/$OUT/pmc_hbm_traffic.txt 2>&1
cd $R
unset MFVI_TUNE_CACHE
echo "done cfg2 whole-iteration profiles"
for spec in "36 16 3 1 256 256" "68 32 3 1 128 128" "132 64 3 1 64 64"; do scripts/dev/layer_traffic.sh $OUT/layer_traffic.txt $spec; done
rm -f $OUT/pmc_sq_up9.txt; scripts/dev/pmc_sq.sh $OUT/pmc_sq_up9.txt 36 16 3 1 256 256
echo "done layer profiles"
python3 scripts/dev/bwdx6_layers.py > $OUT/bwdx6_layers.txt 2>/dev/null
echo "done bwdx6_layers"
# the strip-resident backward-data kernel: in-kernel phase profile and the timing-only variants without its memory operations (built on the box)
scripts/dev/bwdx6s_prof.sh $OUT/bwdx6s_prof.txt > /dev/null 2>&1
scripts/dev/bwdx6s_variants.sh $OUT/bwdx6s_variants.txt > /dev/null 2>&1
echo "done bwdx6s profile / variants"
# bf16x6 kernels (conv_bww_x6.hip, conv_x6.hip): every tiling on the three dominant layers + the 32-wide layer; SQ counters of 68->32 (forward and backward-weight both bf16x6 there)
python3 scripts/dev/x6_layers.py 36 16 256 68 32 128 132 64 64 132 128 32 > $OUT/x6_layers.txt 2>/dev/null
rm -f $OUT/pmc_sq_up7.txt; scripts/dev/pmc_sq.sh $OUT/pmc_sq_up7.txt 68 32 3 1 128 128
echo "done x6 layers"
export MFVI_TUNE_CACHE=$R/$OUT/tunes_cfg2_k1.json
python3 bench.py --k 1 > $OUT/bench_cfg2_k1.json 2> /dev/null
(cd /tmp && rm -rf /tmp/p_k1 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_k1 -- python3 $R/bench.py --k 1 --steps 50 --warmup 3 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1)
cp $(find /tmp/p_k1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_cfg2_k1.csv
unset MFVI_TUNE_CACHE
python3 bench.py --mode dropin --k 1 > $OUT/bench_dropin.json 2> /dev/null
python3 bench.py --mode dropin --k 1 --flat-parameters --no-cpu-baseline > $OUT/bench_dropin_flat.json 2> /dev/null
echo "done k1 / dropin"
for c in cfg1 cfg3 cfg4 cfg5 inp; do
  export MFVI_TUNE_CACHE=$R/$OUT/tunes_$c.json
  python3 bench.py --config $c > $OUT/bench_$c.json 2> /dev/null
  (cd /tmp && rm -rf /tmp/p_$c && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$c -- python3 $R/bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline --no-gpu-baseline > /dev/null 2>&1)
  cp $(find /tmp/p_$c -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$c.csv
  echo "done $c"
done
scripts/dev/graph_bench.sh $OUT/graph_bench.txt > /dev/null 2>&1
echo "done graph bench"
sha256sum mfvi-dip-mia_amd/libmfvi_hip.so > $OUT/lib_sha256.txt
ls -la $OUT
