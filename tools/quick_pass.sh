#!/bin/bash
# Quick GPU pass: GPU tests, the element-wise / BatchNorm1d timing tool, A/B of options inside the step, one bench line without the CPU leg.
cd $GRAFT_REPO_ROOT
o=gpurun_out/${1:-quick}
mkdir -p $o
if [ "$2" != "notests" ]; then timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; grep -E "passed|failed" $o/gpu_tests.log | tail -2; fi
timeout 300 python tools/ew_bench.py 2>&1 | grep -E "maxpool|copy" > $o/ew_bench.txt; cat $o/ew_bench.txt
timeout 300 python tools/ab_step.py 16 3 2>&1 | grep round > $o/ab_step_gat_wide.txt; cat $o/ab_step_gat_wide.txt
timeout 300 python tools/ab_step.py 13 3 0 12 2>&1 | grep round > $o/ab_step_pool12.txt; cat $o/ab_step_pool12.txt
timeout 300 python tools/ab_step.py 13 3 0 16 2>&1 | grep round > $o/ab_step_pool16.txt; cat $o/ab_step_pool16.txt
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 2 > $o/bench_c2.json 2> $o/bench_c2.err; cut -c1-200 $o/bench_c2.json
