#!/bin/bash
# round-5 baseline on this round's box: gpu tests, variant timings, bench line
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5a; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/gpu_tests.log 2>&1; echo "tests rc=$?" 
timeout 200 python tools/wino4_variants_bench.py > $o/w4_variants.txt 2>&1
timeout 200 python tools/wgrad4_bench.py > $o/wgrad4.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > $o/bench_c2.json 2> $o/bench_c2.err
tail -3 $o/gpu_tests.log; cat $o/w4_variants.txt $o/wgrad4.txt; cat $o/bench_c2.json | cut -c1-600
