#!/bin/bash
# Effective shader clock of every big launch INSIDE the bench step (GRBM_GUI_ACTIVE / 8 XCDs / launch duration, the guide's
# recipe) with the F(4x4) forward / data-gradient launches on the split loop and on the f32-MFMA loop: does a launch run at a
# lower clock behind a bf16-MFMA launch?  -> gpurun_out/r5b/pmc_step_clock.txt
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/r5b; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for f32 in 0 1; do
  COVA_W4_F32=$f32 rocprofv3 --pmc GRBM_GUI_ACTIVE -d /tmp/pmcclk_$f32 -- python $root/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-clock-leg --no-ab --sustained-seconds 0 > /tmp/pmcclk_$f32.log 2>&1
  db=$(find /tmp/pmcclk_$f32 -name "*.db" | head -1)
  echo "== COVA_W4_F32=$f32: kernel, launches, mean us, effective GHz (GRBM_GUI_ACTIVE / 8 / duration)"
  [ -n "$db" ] && python - "$db" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection where counter_name = 'GRBM_GUI_ACTIVE' "
                  "group by kernel_name order by sum(duration) desc").fetchall()
for n, c, v, d in rows[:16]:
    n = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:56]
    print("  %-56s %4d %9.1f %6.3f" % (n, c, d / 1e3, v / 8.0 / d))
PY
done > $out/pmc_step_clock.txt 2>&1
cat $out/pmc_step_clock.txt
