cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "tail or f4x4" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -15 > gpurun_out/r3d/t_k.log
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|Warning\|warnings.warn\|m = CoVA\|^$" | tail -12 > gpurun_out/r3d/t.log
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err
COVA_BN_TAIL=0 timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3d/bench_notail.json 2> gpurun_out/r3d/bench_notail.err
