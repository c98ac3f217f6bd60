cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3x
timeout 1500 python -m pytest tests/test_extension_gpu.py tests/test_fullsize_gpu.py -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6 > gpurun_out/r3x/t_e.log
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config 3 > gpurun_out/r3x/bench_c3.json 2> gpurun_out/r3x/bench_c3.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 --config 3 > $GRAFT_REPO_ROOT/gpurun_out/r3x/kt.log 2>&1
db=$(find /tmp/kt -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_step.py $db 2 --order > $GRAFT_REPO_ROOT/gpurun_out/r3x/step_order_c3.txt
