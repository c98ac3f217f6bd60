#!/bin/bash
# ablation builds of the role-split conv1 weight-gradient kernel (C1B_EXTRA=-DWG1R_ABL=<m> tools/c1b_abl_build.sh <m>)
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== WG1R_ABL=$m"
  COVA_HIP_LIB=$PWD/tools/lib/libcova_c1babl_$m.so python tools/conv1_bench.py --time-only 2>&1 | grep "wgrad.*bf16" | tail -1
done
