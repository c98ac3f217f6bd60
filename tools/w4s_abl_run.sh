#!/bin/bash
# one line per ablation build (tools/w4s_abl_build.sh) of the split F(4x4,3x3) main loop
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== W4S_ABL=$m  $(COVA_HIP_LIB=$PWD/tools/lib/libcova_w4sabl_$m.so python tools/w4s_time.py 2>&1 | grep 'split')"
done
