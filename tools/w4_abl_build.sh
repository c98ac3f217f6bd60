#!/bin/bash
# Ablation builds of the F(4x4,3x3) kernel: tools/lib/libcova_w4abl_<mask>.so for every mask given (see W4_ABL in conv_wino4.hip)
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v conv_wino4.o)
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DW4_ABL=$m $W4_EXTRA -c $pkg/csrc/conv_wino4.hip -o $root/tools/lib/obj/w4_$m.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/w4_$m.o -o $root/tools/lib/libcova_w4abl_$m.so && echo built $m ) &
done
wait
