#!/bin/bash
# usage: tools/kres.sh <file.hip> [name filter]   -- VGPR / scratch usage per kernel (gfx950)
f=$1; pat=${2:-.}
hipcc --offload-arch=gfx950 -O3 -std=c++17 -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 \
 | grep -E "error|Function Name|  VGPRs:|ScratchSize|Occupancy" | sed 's/.*remark: //; s/\[-Rpass.*//' \
 | awk '/Function Name/{n=$3} /VGPRs:/{v=$2} /ScratchSize/{s=$3} /Occupancy/{print substr(n,1,90), "vgpr="v, "scratch="s, "occ="$3} /error/{print}' | grep -E "$pat"
