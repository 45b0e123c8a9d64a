set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -40 gpurun_out/pytest.log; exit 1; }
tail -3 gpurun_out/pytest.log
L=gpurun_out/bww.log; : > $L
for shape in "36 16 3 1 256 256" "132 64 3 1 64 64"; do
  for t in "1,4,6" "3,9,1" "3,9,2" "3,9,4"; do
    echo "== $shape tune $t" >> $L; MFVI_AUTOTUNE=0 MFVI_TUNE_W=$t timeout -k 10 120 python scripts/bench_layer.py $shape 2>&1 | grep bwd_weight >> $L
  done
done
cat $L
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all > gpurun_out/bench_at.log 2>&1
tail -1 gpurun_out/bench_at.log | cut -c1-300
