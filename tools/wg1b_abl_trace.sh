#!/bin/bash
# MFMA-loop duration of the conv1 weight-gradient kernel per ablation build (trace builds with -DWG1B_ABL=<m> as lib <m>)
cd "$GRAFT_REPO_ROOT"
for m in "$@"; do
  echo "== WG1B_ABL=$m"
  sed "s/libcova_c1babl_0.so/libcova_c1babl_$m.so/" tools/wg1b_trace.py > /tmp/wg1b_trace_$m.py
  cp /tmp/wg1b_trace_$m.py tools/_wg1b_trace_tmp.py
  python tools/_wg1b_trace_tmp.py 2>&1 | grep -A1 "^wave [04]" | grep tile
  rm -f tools/_wg1b_trace_tmp.py
done
