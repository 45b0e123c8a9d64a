#!/bin/bash
# SQ counters of the row-phase kernels of one layer (three rocprofv3 passes, counters only + kernel trace).
# usage: pmc_rp.sh out_file "cin cout hw" "RP_ONLY spec"
export TMPDIR=/tmp
out=$1; spec=$2; export RP_ONLY=$3; export REPS=3
export MFVI_TUNE_CACHE=/tmp/pmc_rp_tunes.json; rm -f $MFVI_TUNE_CACHE
python3 scripts/dev/rp_layers.py $spec > /dev/null 2>&1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_WAVE32_LDS"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc_$i --output-format csv -- python3 scripts/dev/rp_layers.py $spec > /dev/null 2>&1
  python3 scripts/pmc_summary.py /tmp/pmc_$i conv_rp >> $out
done
