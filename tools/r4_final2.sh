#!/bin/bash
# Round-4 evidence pass, second part (one gpurun call): the bench lines of all configs, kernel trace + HBM counters of the
# bench step at configs[1] / configs[2], SQ counters of the F(4x4) kernels and of the conv1 kernels, the conv1 A/B tables
# (error against fp64, time, ablations, phase traces), the matrix / vector pipe probe, the parity-margin table.
# Results under gpurun_out/r4g; the summaries are copied to profiles/r04_* by hand.  Before: tools/probe/build.sh,
# tools/c1b_abl_build.sh 1 3 4 5 7, C1B_EXTRA=-DC1B_TRACE tools/c1b_abl_build.sh 0 (probe binaries, ablation and trace libraries).
cd $GRAFT_REPO_ROOT
o=gpurun_out/r4g
mkdir -p $o gpurun_out/r4c
timeout 120 ./build/mfma_probe > $o/mfma_probe.txt 2>&1
timeout 120 ./build/mfma32_probe > $o/mfma32_probe.txt 2>&1
timeout 300 python tools/conv1_bench.py > $o/conv1_bench.txt 2>&1
bash tools/c1b_abl_run.sh 1 4 5 3 7 > $o/conv1_fwd_ablation.txt 2>&1
timeout 120 python tools/c1b_trace.py > $o/conv1_fwd_trace.txt 2>&1
timeout 120 python tools/wg1r_trace.py > $o/conv1_wgrad_trace.txt 2>&1
bash tools/pmc_conv1.sh > /dev/null 2>&1; cp gpurun_out/pmc_conv1.txt $o/
bash tools/profile_round.sh r4g/c2 --config 2
bash tools/profile_round.sh r4g/c3 --config 3
PMC_OUT=r4g bash tools/pmc_wino4.sh > /dev/null 2>&1
bash tools/pmc_wgrad4.sh 12 > /dev/null 2>&1; cp gpurun_out/r4c/pmc_wgrad4.txt $o/ 2>/dev/null
cd $GRAFT_REPO_ROOT
timeout 900 python tools/wino4_margin.py $o/r04_parity_margin.txt > $o/margin.log 2>&1
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > $o/bench_c$c.json 2> $o/bench_c$c.err; done
timeout 900 python bench.py > $o/bench_c2.json 2> $o/bench_c2.err
COVA_CONV1_F32=1 timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > $o/bench_c2_conv1f32.json 2> /dev/null
ls $o
