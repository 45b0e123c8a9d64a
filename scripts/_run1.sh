set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -70 gpurun_out/pytest.log; exit 1; }
tail -3 gpurun_out/pytest.log
