set -e
mkdir -p gpurun_out
for d in 0 4; do
MFVI_DBG=$d MFVI_PROF=1 MFVI_AUTOTUNE=0 MFVI_TUNE=1,16,1 timeout -k 10 120 python scripts/bench_layer.py 132 128 3 1 32 32 16 1 > gpurun_out/prof1.log 2>&1
echo "dbg $d"; grep -A8 "^MODE 1 KS 3" gpurun_out/prof1.log | tail -9; grep "bwd_data" gpurun_out/prof1.log | tail -1
done
