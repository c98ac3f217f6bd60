#!/bin/bash
# Quick GPU pass: GPU tests, timing tools, A/B of options inside the step, a profile of the bench step.
cd $GRAFT_REPO_ROOT
o=gpurun_out/${1:-quick}
mkdir -p $o
if [ "$2" != "notests" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -E "passed|failed" $o/gpu_tests.log | tail -2; fi
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $o/gemm_bench.txt; cut -c1-140 $o/gemm_bench.txt
for v in 1 2 3; do timeout 300 python tools/ab_step.py 18 2 0 $v 2>&1 | grep round > $o/ab_step_roipool$v.txt; cut -c260-520 $o/ab_step_roipool$v.txt; done
bash tools/profile_round.sh ${1:-quick}/c2 --config 2 > $o/profile.log 2>&1; tail -3 $o/profile.log
