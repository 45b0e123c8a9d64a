set -e
mkdir -p gpurun_out
for d in 0 1 2 3; do
MFVI_DBG=$d MFVI_PROF=1 MFVI_AUTOTUNE=0 MFVI_TUNE_W=3,9,1 timeout -k 10 120 python scripts/bench_layer.py 36 16 3 1 256 256 16 1 > gpurun_out/prof1.log 2>&1
echo "dbg $d"; grep -A8 "^BWW ks 3" gpurun_out/prof1.log | tail -4
done
