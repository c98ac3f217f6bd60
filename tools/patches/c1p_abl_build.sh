#!/bin/bash
# Ablation builds of conv1 forward with the pool epilogue: tools/lib/libcova_c1p_<C1P_ABL>_<C1B_ABL>.so for every "p:b" pair given
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v "/conv.o")
for m in "$@"; do
  pm=${m%%:*}; bm=${m##*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DC1P_ABL=$pm -DC1B_ABL=$bm $C1B_EXTRA -c $pkg/csrc/conv.hip -o $root/tools/lib/obj/c1p_${pm}_$bm.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/c1p_${pm}_$bm.o -o $root/tools/lib/libcova_c1p_${pm}_$bm.so && echo built $m ) &
done
wait
