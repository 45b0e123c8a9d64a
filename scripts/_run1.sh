set -e
mkdir -p gpurun_out
export MFVI_BENCH_BACKEND=gloo
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > gpurun_out/bench_g2.log 2>&1 || { tail -30 gpurun_out/bench_g2.log; exit 1; }
tail -2 gpurun_out/bench_g2.log | cut -c1-700
