cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout 1200 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -8 > gpurun_out/r3h/t_k.log
timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "not trajectory" 2>&1 | tail -8 > gpurun_out/r3h/t_m.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bits', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
COVA_BN_TAIL=0 timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('notail', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
done > gpurun_out/r3h/ab.log 2>&1
