cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -x 2>&1 | tail -2
python scripts/conv_probe.py 30 24 2>&1 | grep "s2"
run() { echo -n "$1 [$2] : "; ( cd $1 && env $2 python bench.py --steps 20 --warmup 5 --no_roofline --no_cpu_baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' ); echo; }
for i in 1 2 3; do
  run _ab_old "A=1"
  run . "A=1"
done
