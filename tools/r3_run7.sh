cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "f4x4 or tail or fusion" 2>&1 | tail -4 > gpurun_out/r3g/t.log
timeout 300 python tools/wino4_bench.py 2>&1 | grep "F(" > gpurun_out/r3g/w4.log
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.err
