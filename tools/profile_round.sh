#!/bin/bash
# One profiling pass of the bench step on the GPU box: kernel-trace stats + the two HBM counter passes.
#   tools/profile_round.sh <tag> [bench args...]      -> gpurun_out/<tag>_*.txt|json
tag=$1; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/kt_$tag -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-clock-leg --no-ab --sustained-seconds 0 "$@" > $out/${tag}_kt.log 2>&1
db=$(find /tmp/kt_$tag -name "*.db" | head -1)
[ -n "$db" ] && python $root/tools/rocpd_stats.py $db > $out/${tag}_kernel_stats.txt
[ -n "$db" ] && python $root/tools/rocpd_step.py $db 2 --order > $out/${tag}_step_breakdown.txt
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d /tmp/pmc_${tag}_$c -- python $root/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-clock-leg --no-ab --sustained-seconds 0 "$@" > $out/${tag}_pmc_$c.log 2>&1
done
fdb=$(find /tmp/pmc_${tag}_FETCH_SIZE -name "*.db" | head -1)
wdb=$(find /tmp/pmc_${tag}_WRITE_SIZE -name "*.db" | head -1)
pages=$(python - "$@" <<'PY'
import sys
a = sys.argv[1:]
cfg = int(a[a.index("--config") + 1]) if "--config" in a else 2
pages = int(a[a.index("--pages") + 1]) if "--pages" in a else {2: 16, 3: 32, 4: 32, 5: 8}[cfg]
print(pages)
PY
)
[ -n "$fdb" ] && [ -n "$wdb" ] && python $root/tools/hbm_traffic.py $fdb $wdb $pages $out/${tag}_hbm_traffic.json --txt $out/${tag}_pmc_hbm_traffic.txt
