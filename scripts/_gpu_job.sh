cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2000 python -m pytest tests -q -m gpu -x --durations=5 ) > gpurun_out/r3_tests9.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_tests9.log | tail -5
python scripts/wgrad_probe.py 12 2>&1 | grep layer | cut -c1-80
run() { echo -n "$1 [$2] : "; ( cd $1 && env $2 python bench.py --steps 20 --warmup 5 --no_roofline --no_cpu_baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' ); echo; }
for i in 1 2; do
  run _ab_old "A=1"
  run . "A=1"
done
