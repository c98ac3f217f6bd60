#!/bin/bash
# Ablation / trace builds of the split F(4x4,3x3) main loop: tools/lib/libcova_w4sabl_<mask>.so for every mask given (see W4S_ABL
# in csrc/conv_wino4_split.h); W4S_EXTRA adds flags (-DW4S_TRACE ...)
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v "/conv_wino4.o")
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DW4S_ABL=$m $W4S_EXTRA -c $pkg/csrc/conv_wino4.hip -o $root/tools/lib/obj/w4s_$m.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/w4s_$m.o -o $root/tools/lib/libcova_w4sabl_$m.so && echo built $m ) &
done
wait
