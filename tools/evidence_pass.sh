#!/bin/bash
# Round-6 evidence pass (one gpurun call): GPU suite, smoke, the bench lines of all configs (configs[1] with the CPU baseline and the
# clock leg), kernel trace + step cut + HBM counters of the bench step at configs[1] / configs[2], the two-rank gloo DIAGNOSTIC
# line, the 16-page parity report, production sanity (soak, step bit-reproducibility, host-fed rate, poison check, 176-page batch),
# the dispatch-gap probe.  Results under gpurun_out/<tag>; the summaries are copied to profiles/r06_* by hand.
tag=${1:-r6ev}
cd $GRAFT_REPO_ROOT
o=gpurun_out/$tag
mkdir -p $o
timeout 900 python -m pytest tests -m gpu -q > $o/gpu_tests.log 2>&1; tail -1 $o/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1 | tee $o/smoke.txt
timeout 1500 python bench.py > $o/bench_c2.json 2> $o/bench_c2.err
python -c "import json; d=json.load(open('$o/bench_c2.json')); print('configs[1]', d['value'], d['ms_per_step'], d['roofline']['frac'])"
if [ -z "$EVIDENCE_LIGHT" ]; then
bash tools/profile_round.sh $tag/c2 --config 2
bash tools/profile_round.sh $tag/c3 --config 3
cd $GRAFT_REPO_ROOT
for c in 3 5 4; do timeout 600 python bench.py --no-cpu-baseline --no-clock-leg --sustained-seconds 0 --config $c > $o/bench_c$c.json 2> $o/bench_c$c.err; python -c "import json; d=json.load(open('$o/bench_c$c.json')); print('config', $c, d['value'], d['ms_per_step'])"; done
COVA_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-clock-leg --sustained-seconds 0 > $o/bench_gloo2.json 2> $o/bench_gloo2.err
timeout 900 python tests/tools_grad_report_b16.py 16 2>&1 | grep -v amdgpu.ids > $o/grad_parity_1280_b16.txt; tail -3 $o/grad_parity_1280_b16.txt
bash tools/sanity_pass.sh > /dev/null 2>&1; cp gpurun_out/sanity/sanity.txt $o/production_sanity.txt
bash tools/poison_all.sh > $o/poison_check.txt 2>&1; tail -2 $o/poison_check.txt
timeout 600 python tools/large_batch_check.py 2>&1 | grep -v amdgpu.ids > $o/large_batch_check.txt; tail -2 $o/large_batch_check.txt
cd /tmp && export TMPDIR=/tmp; rocprofv3 --kernel-trace -d /tmp/gap -- $GRAFT_REPO_ROOT/build/gap_probe > /dev/null 2>&1; db=$(find /tmp/gap -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/gap_probe.py $db > $GRAFT_REPO_ROOT/$o/gap_probe.txt 2>&1
fi
ls $GRAFT_REPO_ROOT/$o
