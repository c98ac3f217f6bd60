#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4b
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "f4x4_wgrad" 2>&1 | tail -6
python tools/wgrad4_bench.py 02 12 2>&1 | grep prologue
python tools/wgrad4_bench.py 02 12 --emit 2>&1 | grep prologue
python bench.py --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r4b/bench_emit.json 2> gpurun_out/r4b/bench_emit.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4b/bench_emit.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["frac"], {k: v["avg_launch_ms"] for k, v in d["other_kernels"].items() if "wgrad" in k})
PY
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -4
