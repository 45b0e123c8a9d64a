set -e
mkdir -p gpurun_out
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_$i.log 2>&1
tail -1 gpurun_out/bench_$i.log | cut -c140-260
done
