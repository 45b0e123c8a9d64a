set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -70 gpurun_out/pytest.log; exit 1; }
tail -3 gpurun_out/pytest.log
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --profile-all > gpurun_out/bench_at.log 2>&1
tail -1 gpurun_out/bench_at.log | cut -c1-300
MFVI_PROFILE_FULL=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-all > gpurun_out/bench_ops.log 2>&1
grep -E "^op (10|11|13|14|16) " gpurun_out/bench_ops.log
