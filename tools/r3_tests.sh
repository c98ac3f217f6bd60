cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t profiles
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r3t/smoke.log 2>&1
COVA_WRITE_CURVE=$GRAFT_REPO_ROOT/gpurun_out/r3t/trajectory.txt timeout 2400 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|Warning\|warnings.warn\|m = CoVA\|m2 = CoVA\|^$\|^tests/" | tail -30 > gpurun_out/r3t/t.log
timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3t/bench.json 2> gpurun_out/r3t/bench.err
