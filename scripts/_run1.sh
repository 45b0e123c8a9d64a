set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export MFVI_TUNE_CACHE=$R/gpurun_out/tunes.json
timeout -k 10 600 python bench.py > gpurun_out/bench_full.log 2>&1
tail -1 gpurun_out/bench_full.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kt $R/gpurun_out/pmc_f $R/gpurun_out/pmc_w
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1
echo ok
