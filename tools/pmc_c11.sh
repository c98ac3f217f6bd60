#!/bin/bash
# PMC passes over the extension model's step (bench.py --config 3): L1 / L2 request counters per 1x1 kernel
root=$GRAFT_REPO_ROOT
mkdir -p $root/gpurun_out/r3x
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d /tmp/pmc_c11_$i -- python $root/bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > /tmp/pmc_c11_$i.log 2>&1 || tail -3 /tmp/pmc_c11_$i.log
  db=$(find /tmp/pmc_c11_$i -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_pmc.py --raw $db | grep -E "conv1x1|vprod|roipool_fwd"
done > $root/gpurun_out/r3x/pmc_c11.txt 2>&1
