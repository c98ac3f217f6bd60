# Round-3 evidence pass (one gpurun call): kernel trace + HBM counters of the bench step at configs[1] and configs[2],
# SQ counters of the F(4x4) / F(2x2) kernels, the parity-margin table, the bench lines of all configs.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f gpurun_out/r3b gpurun_out/r3m
bash tools/profile_round.sh r3f/c2 --config 2
bash tools/profile_round.sh r3f/c3 --config 3
bash tools/pmc_wino4.sh
cd $GRAFT_REPO_ROOT
timeout 900 python tools/wino4_margin.py gpurun_out/r3m/r03_wino4_margin.txt > gpurun_out/r3m/margin.log 2>&1
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > gpurun_out/r3m/bench_c$c.json 2> gpurun_out/r3m/bench_c$c.err; done
timeout 900 python bench.py > gpurun_out/r3f/bench_c2.json 2> gpurun_out/r3f/bench_c2.err
