cd $GRAFT_REPO_ROOT
run() { echo -n "[$1] : "; ( env $1 timeout 300 python bench.py --steps 20 --warmup 5 --no_roofline --no_cpu_baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' ); echo; }
for i in 1 2 3; do
  run "FD_REFLECT_RING=0"
  run "FD_REFLECT_RING=1"
done
