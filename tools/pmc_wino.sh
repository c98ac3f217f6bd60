#!/bin/bash
# PMC passes over one Winograd conv configuration: tools/pmc_wino.sh <mode> ; output in gpurun_out/pmc_<mode>.txt
mode=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_${mode}_$i -- python $root/tools/wino_profile_target.py $mode > /tmp/pmc_${mode}_$i.log 2>&1 || tail -3 /tmp/pmc_${mode}_$i.log
  db=$(find /tmp/pmc_${mode}_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_pmc.py --raw $db | grep -E "wino_kernel|conv1_|wgrad_v2" 
done > $root/gpurun_out/pmc_$mode.txt 2>&1
