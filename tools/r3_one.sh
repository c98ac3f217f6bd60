cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3o
timeout 900 python -m pytest tests/test_syncbn_gpu.py tests/test_kernels_gpu.py -q -x -k "syncbn or batchnorm1d or rccl or local_batchnorm" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -12 > gpurun_out/r3o/a.log
python tools/tail_ab.py > gpurun_out/r3o/ab.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3o/bench.json 2> gpurun_out/r3o/bench.err
