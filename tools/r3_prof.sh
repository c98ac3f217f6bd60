cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "tail" 2>&1 | tail -3 > gpurun_out/r3e/t_k.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r3e/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_step.py $db 2 --order > $GRAFT_REPO_ROOT/gpurun_out/r3e/step_order.txt
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $db > $GRAFT_REPO_ROOT/gpurun_out/r3e/kernel_stats.txt
