cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
timeout 900 python -m pytest tests/test_syncbn_gpu.py -q -x 2>&1 | grep -v "Warning\|warnings.warn\|^\[W" | tail -40 > gpurun_out/r3t/one.log
