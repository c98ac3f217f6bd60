#!/bin/bash
# PMC passes over the F(4x4,3x3) kernel (tools/w4s_time.py --f32: the five step variants on the f32 and on the split main loop): output in gpurun_out/${PMC_OUT:-r3b}/pmc_wino4.txt
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_w4_$i -- python $root/tools/w4s_time.py --f32 > /tmp/pmc_w4_$i.log 2>&1 || tail -3 /tmp/pmc_w4_$i.log
  db=$(find /tmp/pmc_w4_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_pmc.py --raw $db | grep -E "wino4s_kernel|wino4_kernel|c64_wino_kernel"
done > $root/gpurun_out/${PMC_OUT:-r3b}/pmc_wino4.txt 2>&1
