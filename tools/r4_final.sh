#!/bin/bash
# Round-4 evidence pass (one gpurun call): kernel trace + HBM counters of the bench step at configs[1] and configs[2], SQ
# counters of the F(4x4) forward / data-gradient and weight-gradient kernels, the parity-margin table, the bench lines of
# all configs.  Results under gpurun_out/r4f, r4m; the summaries are copied to profiles/r04_* by hand.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f gpurun_out/r4m gpurun_out/r4c
bash tools/profile_round.sh r4f/c2 --config 2
bash tools/profile_round.sh r4f/c3 --config 3
PMC_OUT=r4f bash tools/pmc_wino4.sh
bash tools/pmc_wgrad4.sh 12 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python tools/wino4_margin.py gpurun_out/r4m/r04_parity_margin.txt > gpurun_out/r4m/margin.log 2>&1
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > gpurun_out/r4m/bench_c$c.json 2> gpurun_out/r4m/bench_c$c.err; done
timeout 900 python bench.py > gpurun_out/r4f/bench_c2.json 2> gpurun_out/r4f/bench_c2.err
ls gpurun_out/r4f gpurun_out/r4m gpurun_out/r4c
