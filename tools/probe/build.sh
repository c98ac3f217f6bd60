#!/bin/bash
# builds the stand-alone probes into build/ (git-ignored; travels to the GPU box with the gpurun snapshot)
root=$(cd $(dirname $0)/../.. && pwd)
mkdir -p $root/build
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize $root/tools/probe/mfma_probe.hip -o $root/build/mfma_probe 2>/dev/null && echo built mfma_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/probe/mfma32_probe.hip -o $root/build/mfma32_probe 2>/dev/null && echo built mfma32_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize $root/tools/probe/mfma16_probe.hip -o $root/build/mfma16_probe 2>/dev/null && echo built mfma16_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/probe/mfma_round_probe.hip -o $root/build/mfma_round_probe 2>/dev/null && echo built mfma_round_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/probe/mfma_round_probe2.hip -o $root/build/mfma_round_probe2 2>/dev/null && echo built mfma_round_probe2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/probe/gap_probe.hip -o $root/build/gap_probe 2>/dev/null && echo built gap_probe
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 $root/tools/probe/dma_probe.hip -o $root/build/dma_probe 2>/dev/null && echo built dma_probe
