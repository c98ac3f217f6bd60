cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
for v in $W4_LIBS; do
  if [ "$v" = "base" ]; then L=""; else L="COVA_HIP_LIB=$GRAFT_REPO_ROOT/tools/lib/libcova_w4abl_$v.so"; fi
  env $L timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r3f/bench_$v.json 2> gpurun_out/r3f/bench_$v.err
done
