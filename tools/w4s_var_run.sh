#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== $m  $(COVA_HIP_LIB=$PWD/tools/lib/libcova_w4svar_$m.so python tools/w4s_time.py 2>&1 | grep 'split')"
done
