cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_convstack.py -q -m gpu -k "reflect or conv2d_fwd_bwd or routing" ) > gpurun_out/r3_t26.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t26.log | tail -8
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'])"; }
for i in 1 2 3; do
run FD_REFLECT_WINO_PADDED=1
run FD_REFLECT_WINO_PADDED=0
done
