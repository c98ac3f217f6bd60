#!/bin/bash
# cova_set_option(21, .) (XCD-contiguous tile walk of the conv1 kernels): its test, the interleaved A/B inside the step, and the
# FETCH_SIZE of the two conv1 launches in both modes.   -> gpurun_out/xcdwalk
cd $GRAFT_REPO_ROOT
o=$GRAFT_REPO_ROOT/gpurun_out/xcdwalk; mkdir -p $o
timeout 300 python -m pytest tests/test_kernels_gpu.py -q -k "conv1" > $o/tests.log 2>&1; tail -1 $o/tests.log
timeout 300 python tools/ab_step.py 21 4 2>&1 | grep round > $o/ab_step_xcdwalk.txt; cat $o/ab_step_xcdwalk.txt | cut -c1-260
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  COVA_OPTION_21=$v rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_xw$v -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $o/pmc_$v.log 2>&1
  db=$(find /tmp/pmc_xw$v -name "*.db" | head -1)
  [ -n "$db" ] && python $GRAFT_REPO_ROOT/tools/rocpd_pmc.py $db 2>/dev/null | grep -E "conv1_7x7_bf3|conv1_wgrad_rs|bn_relu_maxpool" > $o/fetch_$v.txt
  cat $o/fetch_$v.txt | cut -c1-200
done
