#!/bin/bash
# Round-5 evidence pass (one gpurun call): the bench lines of all configs, kernel trace + step cut + HBM counters of the bench step
# at configs[1] / configs[2], SQ counters of the F(4x4) kernels (split and f32 main loops), the interleaved A/B of the two loops,
# the matrix-pipe probes, the parity-margin table, the two-rank gloo DIAGNOSTIC line, the error class of the bf16-pipe 1x1 products and
# the full-page parity report of the ResNet-50 extension.  Results under gpurun_out/r5g; the summaries
# are copied to profiles/r05_* by hand.  Before: tools/probe/build.sh.
cd $GRAFT_REPO_ROOT
o=gpurun_out/r5g
mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q > $o/gpu_tests.log 2>&1; tail -1 $o/gpu_tests.log
# the latency-bound launches: element-wise / pool launch shapes, BatchNorm1d forms, GEMM yardstick, A/B inside the step
timeout 300 python tools/ew_bench.py 2>&1 | grep -v amdgpu.ids > $o/ew_bench.txt
timeout 300 python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids > $o/gemm_bench.txt
timeout 120 python tools/gat_wide_debug.py 2>&1 | grep "^D " > $o/gat_wide_debug.txt
timeout 300 python tools/ab_step.py 13 3 0 12 2>&1 | grep round > $o/ab_step_pool.txt
timeout 300 python tools/ab_step.py 16 3 2>&1 | grep round > $o/ab_step_gat_wide.txt
timeout 300 python tools/ab_step.py 18 3 0 3 2>&1 | grep round > $o/ab_step_roipool.txt
timeout 300 python tools/ab_step.py 14 3 2>&1 | grep round > $o/ab_step_bn1d.txt
timeout 300 python tools/ab_step.py 9 4 2>&1 | grep round > $o/ab_step_wino4.txt
timeout 300 python tools/ab_step.py 7 3 2>&1 | grep round > $o/ab_step_conv1.txt
if [ -z "$EVIDENCE_LIGHT" ]; then      # the probes of the matrix pipe and the split kernels (unchanged code: skipped in a light pass)
timeout 120 ./build/mfma16_probe > $o/mfma16_probe.txt 2>&1
timeout 60 ./build/mfma_round_probe2 > $o/mfma_round_probe2.txt 2>&1
timeout 300 python tools/w4s_bias.py 2>&1 | grep -v amdgpu.ids > $o/w4s_bias.txt
timeout 300 python tools/w4s_time.py --f32 2>&1 | grep loop > $o/w4s_time.txt
timeout 300 python tools/w4s_check.py 2>&1 | grep -v amdgpu.ids > $o/w4s_check.txt
timeout 300 python tools/ab_step.py 11 2 2>&1 | grep round > $o/ab_step_sgemm.txt
timeout 300 python tools/ab_step.py 12 2 2>&1 | grep round > $o/ab_step_conv1w4.txt
timeout 300 python tools/c11_error.py 2>&1 | grep -v amdgpu.ids > $o/c11_error.txt
timeout 900 python tests/tools_grad_report_r50.py 4 2>&1 | grep -v amdgpu.ids > $o/grad_parity_1280_r50.txt
PMC_OUT=r5g bash tools/pmc_wino4.sh > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python tools/wino4_margin.py $o/r05_parity_margin.txt > $o/margin.log 2>&1
fi
timeout 900 python tests/tools_grad_report_b16.py 16 2>&1 | grep -v amdgpu.ids > $o/grad_parity_1280_b16.txt
bash tools/profile_round.sh r5g/c2 --config 2
bash tools/profile_round.sh r5g/c3 --config 3
cd $GRAFT_REPO_ROOT
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > $o/bench_c$c.json 2> $o/bench_c$c.err; done
COVA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --sustained-seconds 0 > $o/bench_gloo2.json 2> $o/bench_gloo2.err
timeout 1200 python bench.py > $o/bench_c2.json 2> $o/bench_c2.err
ls $o
