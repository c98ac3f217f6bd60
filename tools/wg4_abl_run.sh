#!/bin/bash
# one line per ablation build (tools/wg4_abl_build.sh): the four prologue variants of the F(4x4) weight-gradient kernel
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== WG4_ABL=$m"
  COVA_HIP_LIB=$PWD/tools/lib/libcova_wg4abl_$m.so python tools/wgrad4_bench.py 2>&1 | grep prologue | awk '{print $2, $3, $5}' | tr '\n' ';'
  echo
done
