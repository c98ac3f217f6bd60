#!/bin/bash
# Quick GPU pass: GPU tests, timing tools, A/B of options inside the step.
cd $GRAFT_REPO_ROOT
o=gpurun_out/${1:-quick}
mkdir -p $o
if [ "$2" != "notests" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -E "passed|failed" $o/gpu_tests.log | tail -2; fi
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $o/gemm_bench.txt; cut -c1-50,140-260 $o/gemm_bench.txt
timeout 300 python tools/ab_step.py 19 3 2>&1 | grep round > $o/ab_step_sgemm_pf2.txt; cut -c1-40,300-520 $o/ab_step_sgemm_pf2.txt
timeout 300 python tools/ab_step.py 20 3 2 4 2>&1 | grep round > $o/ab_step_roipool_bwd.txt; cut -c1-40,300-520 $o/ab_step_roipool_bwd.txt
