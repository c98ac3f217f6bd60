cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
for m in new stag1 stag2 stag3 stag4 new; do
  if [ $m = new ]; then lib=""; else lib=$GRAFT_REPO_ROOT/tools/lib/libcova_$m.so; fi
  echo "$m: $(COVA_HIP_LIB=$lib timeout 120 python tools/wino4_variants_bench.py 2>&1 | grep ' ms :' )"
done > gpurun_out/r3a/abl4.log 2>&1
