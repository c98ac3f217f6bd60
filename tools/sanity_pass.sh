#!/bin/bash
# Production sanity on the final code (one gpurun call): soak (flat memory, steady step time, finite loss), bit-reproducibility of
# whole train steps at the benchmark sizes, the host-fed (PCIe-inclusive) rate.   -> gpurun_out/sanity/sanity.txt
cd $GRAFT_REPO_ROOT
o=gpurun_out/sanity; mkdir -p $o
{
echo "# STEPS=2000 python tools/soak.py"; STEPS=2000 timeout 300 python tools/soak.py 2>&1 | grep -v amdgpu.ids
echo; echo "# python tools/repro_check.py"; timeout 300 python tools/repro_check.py 2>&1 | grep -v amdgpu.ids
echo; echo "# STEPS=100 python tools/pcie_inclusive.py"; STEPS=100 timeout 300 python tools/pcie_inclusive.py 2>&1 | grep -v amdgpu.ids
} > $o/sanity.txt
cat $o/sanity.txt
