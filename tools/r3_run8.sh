cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -8 > gpurun_out/r3h/t_k.log
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "not trajectory" 2>&1 | tail -8 > gpurun_out/r3h/t_m.log
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/r3h/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_step.py $db 2 --order > $GRAFT_REPO_ROOT/gpurun_out/r3h/step_order.txt
