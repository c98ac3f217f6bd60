#!/bin/bash
# Quick GPU pass: the GPU tests, then (optional) the kernel trace of the bench step with a second library build for a same-box A/B
# (the step cut is step 3 of the trace = the SECOND timed step: the first one carries bench.py's HIP events, 6 us of dispatch each)
# of two builds:  COVA_AB_LIB=<path to the other libcova_hip.so> tools/quick_pass.sh <tag>
cd $GRAFT_REPO_ROOT
o=gpurun_out/${1:-quick}
mkdir -p $o
root=$GRAFT_REPO_ROOT
if [ "$2" != "notests" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -E "passed|failed" $o/gpu_tests.log | tail -2; fi
cd /tmp && export TMPDIR=/tmp
for tag in prev new prev2 new2; do
  if [ "${tag:0:4}" = "prev" ]; then [ -z "$COVA_AB_LIB" ] && continue; export COVA_HIP_LIB=$COVA_AB_LIB; else unset COVA_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -- python $root/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-clock-leg --no-ab --sustained-seconds 0 > $root/$o/${tag}_kt.log 2>&1
  db=$(find /tmp/kt_$tag -name "*.db" | head -1)
  [ -n "$db" ] && python $root/tools/rocpd_step.py $db 3 --order > $root/$o/${tag}_step_breakdown.txt
  head -1 $root/$o/${tag}_step_breakdown.txt
done
