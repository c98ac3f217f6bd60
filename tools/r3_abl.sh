cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
for m in 0 32 512 256 288 4 36; do
  if [ $m = 0 ]; then lib=""; else lib=$GRAFT_REPO_ROOT/tools/lib/libcova_w4abl_$m.so; fi
  echo "mask $m: $(COVA_HIP_LIB=$lib timeout 120 python tools/wino4_bench.py 2>&1 | grep 'F(' )"
done > gpurun_out/r3a/abl.log 2>&1
