#!/bin/bash
# HBM read bytes of the backward-weight kernel of one layer for several tilings (rocprofv3 --pmc FETCH_SIZE, FETCH doubled)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in "1,4,1" "1,9,1" "3,9,1" "3,9,4" "3,8,1" "2,9,1"; do
  rm -rf gpurun_out/bf
  MFVI_TUNE_W=$t rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/bf -- python3 scripts/bench_layer.py $1 $2 3 1 $3 $3 16 3 > gpurun_out/bf.log 2>&1
  f=$(ls gpurun_out/bf/*/*counter_collection.csv | head -1)
  python3 - "$f" "$t" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "bww" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        agg[(r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in agg.items():
    v.sort(); print(sys.argv[2], k, "rd %.1f MB" % (2 * v[len(v) // 2] * 1024 / 1e6), "n", len(v))
PY
  grep bwd_weight gpurun_out/bf.log
done
rm -rf gpurun_out/bf
