cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout 900 python tools/wino4_margin.py gpurun_out/r3m/r03_wino4_margin.txt > gpurun_out/r3m/margin.log 2>&1
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > gpurun_out/r3m/bench_c$c.json 2> gpurun_out/r3m/bench_c$c.err; done
