cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3ab
for i in 1 2 3; do
  for v in new old; do
    if [ $v = new ]; then lib=""; else lib=$GRAFT_REPO_ROOT/tools/lib/libcova_old.so; fi
    COVA_HIP_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --sustained-seconds 0 --steps 60 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done > gpurun_out/r3ab/ab.log 2>&1
