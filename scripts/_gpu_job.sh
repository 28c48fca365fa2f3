cd $GRAFT_REPO_ROOT
for m in 64 32 16; do echo "== FD_WINO_MIN_M=$m"; FD_WINO_MIN_M=$m python scripts/conv_probe.py 30 12 2>&1 | grep "refl" | cut -c1-70; done
run() { echo -n "[$1] : "; ( env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no_roofline --no_cpu_baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' ); echo; }
for i in 1 2; do
  run "FD_WINO_MIN_M=64"
  run "FD_WINO_MIN_M=32"
  run "FD_WINO_MIN_M=16"
done
