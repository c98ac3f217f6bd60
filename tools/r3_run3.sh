cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
for m in 1 2 3 4 7 8 16 24 32 28 31 63; do
  echo "== abl $m" ; COVA_HIP_LIB=$GRAFT_REPO_ROOT/tools/lib/libcova_w4abl_$m.so timeout 120 python tools/wino4_bench.py 2>&1 | grep "F(2x2"
done > gpurun_out/r3b/abl.log 2>&1
