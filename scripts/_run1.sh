set -e
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
export MFVI_TUNE_CACHE=$R/gpurun_out/tunes.json
rm -f $MFVI_TUNE_CACHE
timeout -k 10 600 python bench.py > gpurun_out/bench_full.log 2>&1
tail -1 gpurun_out/bench_full.log | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/kt --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_f --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_w.log 2>&1
cd $R
MFVI_PROFILE_FULL=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --profile-all > gpurun_out/bench_ops.log 2>&1
python - <<'PY' > gpurun_out/tunes.log 2>&1
import sys; sys.path.insert(0,'.')
import torch
from mfvi_dip_mia_amd.engine import ElboEngine
e=ElboEngine(256,256,K=16)
for i,tu in e.plan.tunes().items():
    o=e.prog.ops[i]; print(i,o['ksize'],o['stride'],e.prog.tensors[o['in0']]['C'],'->',e.prog.tensors[o['out']]['C'],'@',e.prog.tensors[o['out']]['H'],*tu)
PY
echo ok
