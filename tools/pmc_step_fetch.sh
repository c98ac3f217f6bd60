#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the wgrad4 and wino4 launches inside the bench step with the pair pacing on and off
# -> gpurun_out/r5b/pmc_wg4_sync.txt
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r5b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for sync in 1 0; do
  for c in FETCH_SIZE; do
    COVA_WG4_PAIR_SYNC=$sync rocprofv3 --pmc $c -d /tmp/pmcs_${sync}_$c -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > /tmp/pmcs_${sync}_$c.log 2>&1
    db=$(find /tmp/pmcs_${sync}_$c -name "*.db" | head -1)
    echo "== pair sync $sync $c"
    [ -n "$db" ] && python $root/tools/rocpd_pmc.py $db | grep -E "wgrad4_kernel|wino4s_kernel"
  done
done > $out/pmc_wg4_sync.txt 2>&1
cat $out/pmc_wg4_sync.txt
