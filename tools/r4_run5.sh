#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4c
timeout 1200 python -m pytest tests/test_extension_gpu.py -m gpu -x -q 2>&1 | tail -3
for c in 3 5; do
  timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > gpurun_out/r4c/bench_c$c.json 2> gpurun_out/r4c/bench_c$c.err
  COVA_WGRAD4=0 timeout 600 python bench.py --no-cpu-baseline --sustained-seconds 0 --config $c > gpurun_out/r4c/bench_c${c}_wg2.json 2> /dev/null
done
python - <<'PY'
import json
for n in ("c3", "c3_wg2", "c5", "c5_wg2"):
    try:
        d = json.load(open("gpurun_out/r4c/bench_%s.json" % n)); print(n, d["value"], d["ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
