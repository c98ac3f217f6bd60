#!/bin/bash
# Ablation builds of the bf16-split conv1 kernel: tools/lib/libcova_c1babl_<mask>.so for every mask given (see C1B_ABL)
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v "/conv.o")
for m in "$@"; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DC1B_ABL=$m $C1B_EXTRA -c $pkg/csrc/conv.hip -o $root/tools/lib/obj/c1b_$m.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/c1b_$m.o -o $root/tools/lib/libcova_c1babl_$m.so && echo built $m ) &
done
wait
