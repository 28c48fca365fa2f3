cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do
run FD_STACK_DIRECT=1
run FD_STACK_DIRECT=0
done
