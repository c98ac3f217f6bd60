cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "f4x4" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -5 > gpurun_out/r3b/t.log
timeout 300 python tools/wino4_bench.py > gpurun_out/r3b/bench.log 2>&1
for m in $W4_MASKS; do
  echo "== abl $m" ; COVA_HIP_LIB=$GRAFT_REPO_ROOT/tools/lib/libcova_w4abl_$m.so timeout 120 python tools/wino4_bench.py 2>&1 | grep "F(2x2"
done > gpurun_out/r3b/abl.log 2>&1
