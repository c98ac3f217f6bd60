#!/bin/bash
# one line per ablation build (tools/c1b_abl_build.sh) of the conv1 forward kernel
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== C1B_ABL=$m"
  COVA_HIP_LIB=$PWD/tools/lib/libcova_c1babl_$m.so python tools/conv1_bench.py --time-only 2>&1 | grep "bf16 split"
done
