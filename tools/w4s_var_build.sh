#!/bin/bash
# Variant builds of the split F(4x4,3x3) main loop: tools/lib/libcova_w4svar_<name>.so; arguments "name:flags"
root=$(cd $(dirname $0)/.. && pwd)
pkg=$root/cova-web-object-detection_amd
mkdir -p $root/tools/lib/obj
others=$(ls $pkg/lib/obj/*.o | grep -v "/conv_wino4.o")
for a in "$@"; do
  name=${a%%:*}; flags=${a#*:}
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $flags -c $pkg/csrc/conv_wino4.hip -o $root/tools/lib/obj/w4sv_$name.o 2>/dev/null &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $root/tools/lib/obj/w4sv_$name.o -o $root/tools/lib/libcova_w4svar_$name.so && echo built $name ) &
done
wait
