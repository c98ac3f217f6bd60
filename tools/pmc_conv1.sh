#!/bin/bash
# PMC passes over the conv1 forward kernels (tools/conv1_bench.py): output in gpurun_out/pmc_conv1.txt
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_c1_$i -- python $root/tools/conv1_bench.py --time-only > /tmp/pmc_c1_$i.log 2>&1 || tail -3 /tmp/pmc_c1_$i.log
  db=$(find /tmp/pmc_c1_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_pmc.py --raw $db | grep -E "conv1_"
done > $out/pmc_conv1.txt 2>&1
cat $out/pmc_conv1.txt
