#!/bin/bash
# VGPR / spill report of a few conv_mfma_kernel instantiations without compiling the ~300 the launcher needs (30 s instead of 5 min):
#   scripts/dev/regs_conv_mfma.sh "3,1,2,8,1,true,false,false,false,true,true" "3,1,4,8,1,false,false,false,false,true,true" ...
# template arguments: KS, STRIDE, MF, TH, MODE, WS, FLAT, BIGC, PH, FF, REM
cd "$(dirname "$0")/../.." || exit 1
T=mfvi-dip-mia_amd/csrc/_regs_tmp.hip
{ echo '#define MFVI_KERNEL_ONLY'; echo '#include "conv_mfma.hip"'; echo 'namespace {'
  for a in "$@"; do echo "template __global__ void conv_mfma_kernel<$a>(MfmaArgs);"; done
  echo '}'; } > $T
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function $EXTRA -Rpass-analysis=kernel-resource-usage -c $T -o /tmp/_regs_tmp.o 2>&1 | python3 -c "
import re,sys
t=sys.stdin.read()
errs=[l for l in t.split('\n') if 'error' in l]
print('\n'.join(errs[:10]))
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name=b.split()[0]
    g=lambda k: (re.search(k+r': (\d+)', b) or [0,'-'])[1]
    m=re.search(r'conv_mfma_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)', name)
    print(m.groups() if m else name[:60], 'VGPRs', g('VGPRs'), 'spill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'), 'occ', g(r'Occupancy \[waves/SIMD\]'))
"
rm -f $T
