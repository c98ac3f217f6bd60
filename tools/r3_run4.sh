cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -25 > gpurun_out/r3c/t.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
COVA_WINO4=0 timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r3c/bench_w2.json 2> gpurun_out/r3c/bench_w2.err
