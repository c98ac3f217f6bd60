#!/bin/bash
# Quick GPU pass: GPU tests, the element-wise / BatchNorm1d timing tool, the A/B of an option inside the step, one bench line without the CPU leg.
cd $GRAFT_REPO_ROOT
o=gpurun_out/${1:-quick}
mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; tail -3 $o/gpu_tests.log
timeout 300 python tools/ew_bench.py 2>&1 | grep -v amdgpu.ids > $o/ew_bench.txt; cat $o/ew_bench.txt
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $o/gemm_bench.txt; cat $o/gemm_bench.txt
timeout 300 python tools/ab_step.py 15 3 2>&1 | grep round > $o/ab_step_sgemm_direct.txt; cat $o/ab_step_sgemm_direct.txt
timeout 300 python tools/ab_step.py 14 3 2>&1 | grep round > $o/ab_step_bn1d.txt; cat $o/ab_step_bn1d.txt
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 2 > $o/bench_c2.json 2> $o/bench_c2.err; cut -c1-400 $o/bench_c2.json
