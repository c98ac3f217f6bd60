#!/bin/bash
# Ablation builds of the F(4x4,3x3) weight-gradient kernel: tools/lib/libcova_wg4abl_<mask>.so for every mask given (see WG4_ABL)
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v conv_wgrad4.o)
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DWG4_ABL=$m $WG4_EXTRA -c $pkg/csrc/conv_wgrad4.hip -o $root/tools/lib/obj/wg4_$m.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/wg4_$m.o -o $root/tools/lib/libcova_wg4abl_$m.so && echo built $m ) &
done
wait
