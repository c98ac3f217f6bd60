cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3t
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|Warning\|warnings.warn\|m = CoVA\|m2 = CoVA\|^$\|^tests/" | tail -12 > gpurun_out/r3t/t.log
