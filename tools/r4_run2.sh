#!/bin/bash
# wgrad4 bring-up: kernel tests first, then the whole GPU suite and the bench (F(4x4) wgrad on / off)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "f4x4_wgrad" 2>&1 | tail -15 > gpurun_out/r4b/wg4_tests.log
cat gpurun_out/r4b/wg4_tests.log | tail -6
if grep -q "failed\|error" gpurun_out/r4b/wg4_tests.log; then exit 1; fi
python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r4b/bench_wg4.json 2> gpurun_out/r4b/bench_wg4.err
COVA_WGRAD4=0 python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r4b/bench_wg2.json 2> gpurun_out/r4b/bench_wg2.err
python - <<'PY'
import json
for n in ("wg4", "wg2"):
    try:
        d = json.load(open("gpurun_out/r4b/bench_%s.json" % n))
        print(n, d["value"], d["ms_per_step"], {k: v["avg_launch_ms"] for k, v in d["other_kernels"].items() if "wgrad" in k})
    except Exception as e:
        print(n, "failed", e)
PY
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r4b/tests.log
tail -4 gpurun_out/r4b/tests.log
