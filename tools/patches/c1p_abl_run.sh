#!/bin/bash
# one line per ablation build of tools/c1p_abl_build.sh
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  pm=${m%%:*}; bm=${m##*:}
  echo "== C1P_ABL=$pm C1B_ABL=$bm: $(COVA_HIP_LIB=$PWD/tools/lib/libcova_c1p_${pm}_$bm.so python tools/c1p_bench.py 2>&1 | grep conv1)"
done
