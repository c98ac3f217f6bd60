#!/bin/bash
# round-4 first GPU pass: GPU tests, bench, N=2 code-path checks
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r4a/tests.log
python bench.py --no-cpu-baseline > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err
# N = 2 on one GPU: the preflight must answer with an error line, not a hang
timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 0 > gpurun_out/r4a/bench_n2_preflight.json 2> gpurun_out/r4a/bench_n2_preflight.err
echo "rc $?" >> gpurun_out/r4a/bench_n2_preflight.json
# ... and the shared-device diagnostic run of the whole multi-rank code path (gloo)
COVA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --sustained-seconds 1 > gpurun_out/r4a/bench_n2_gloo.json 2> gpurun_out/r4a/bench_n2_gloo.err
echo "rc $?" >> gpurun_out/r4a/bench_n2_gloo.json
tail -5 gpurun_out/r4a/tests.log
cat gpurun_out/r4a/bench.json | cut -c1-600
