#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== lib $m"
  COVA_HIP_LIB=$PWD/tools/lib/libcova_c1babl_$m.so python tools/conv1_bench.py --time-only 2>&1 | grep "wgrad.*bf16"
done
