# Runs tools/wino4_variants_bench.py against the product library and against ablation builds of the F(4x4,3x3) kernel
# (tools/w4_abl_build.sh <mask>...; masks: see W4_ABL in csrc/conv_wino4.hip).  On the GPU box:
#   bash tools/w4_abl_build.sh 32 8 4 3 256 && gpurun -- 'bash tools/w4_abl_run.sh 0 32 8 4 3 256'
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/w4abl
for m in "$@"; do
  if [ $m = 0 ]; then lib=""; else lib=$GRAFT_REPO_ROOT/tools/lib/libcova_w4abl_$m.so; fi
  echo "mask $m: $(COVA_HIP_LIB=$lib timeout 120 python tools/wino4_variants_bench.py 2>&1 | grep ' ms :' )"
done > gpurun_out/w4abl/abl.log 2>&1
