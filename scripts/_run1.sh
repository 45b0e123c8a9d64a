set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1 || { tail -30 gpurun_out/pytest.log; exit 1; }
tail -3 gpurun_out/pytest.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-all > gpurun_out/bench_at.log 2>&1
tail -5 gpurun_out/bench_at.log
MFVI_AUTOTUNE=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_noat.log 2>&1
tail -1 gpurun_out/bench_noat.log
python - <<'PY' > gpurun_out/tunes.log 2>&1
import sys, time; sys.path.insert(0,'.')
import torch, mfvi_dip_mia_amd as M
from mfvi_dip_mia_amd.engine import ElboEngine
t=time.time(); e=ElboEngine(256,256,K=16); torch.cuda.synchronize(); print('init s',time.time()-t)
for i,(f,b) in e.plan.tunes().items():
    o=e.prog.ops[i]; print(i,o['ksize'],o['stride'],e.prog.tensors[o['in0']]['C'],'->',e.prog.tensors[o['out']]['C'],'@',e.prog.tensors[o['out']]['H'],f,b)
PY
cat gpurun_out/tunes.log
