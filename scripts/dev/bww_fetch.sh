#!/bin/bash
# HBM read bytes of the backward-weight kernel of ONE layer for several tilings (rocprofv3 --pmc FETCH_SIZE, doubled: gfx950).
# bench_layer.py's plan is conv0 (1x1) -> conv under test -> conv2 (1x1): per backward the bww dispatches come in the order
# conv2, test, conv0, so the layer under test is every 3k+1-th bww dispatch.   usage: bww_fetch.sh cin cout size "nb,w,tb" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cin=$1; cout=$2; size=$3; shift 3
for t in "$@"; do
  rm -rf gpurun_out/bf
  MFVI_TUNE_W=$t rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/bf -- python3 scripts/bench_layer.py $cin $cout 3 1 $size $size 16 3 > gpurun_out/bf.log 2>&1
  f=$(ls gpurun_out/bf/*/*counter_collection.csv | head -1)
  python3 - "$f" "$t" <<'PY'
import csv, sys
rows = [(int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["Grid_Size"]), float(r["Counter_Value"])) for r in csv.DictReader(open(sys.argv[1]))
        if "bww" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort()
test = rows[1::3]
v = sorted(x[3] for x in test)
print("tiling %-8s grid %7d threads  HBM read %.1f MB per launch (n=%d)" % (sys.argv[2], test[0][2], 2 * v[len(v) // 2] * 1024 / 1e6, len(v)))
PY
done
rm -rf gpurun_out/bf
