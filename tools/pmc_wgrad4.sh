#!/bin/bash
# PMC passes over the F(4x4,3x3) weight-gradient kernel (tools/wgrad4_bench.py 12 = both operand prologues):
# output in gpurun_out/r4c/pmc_wgrad4.txt
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r4c
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_wg4_$i -- python $root/tools/wgrad4_bench.py ${1:-12} > /tmp/pmc_wg4_$i.log 2>&1 || tail -3 /tmp/pmc_wg4_$i.log
  db=$(find /tmp/pmc_wg4_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_pmc.py --raw $db | grep -E "wgrad"
done > $out/pmc_wgrad4.txt 2>&1
cat $out/pmc_wgrad4.txt
