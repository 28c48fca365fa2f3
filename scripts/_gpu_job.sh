cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -k 10 800 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_dp.py -q -m gpu -x ) > gpurun_out/r3_t35.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t35.log | tail -5
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3), d['param_checksum'], d['optimizer_steps_run'])"; }
for i in 1 2 3; do
run FD_ADAM_SPLIT=1
run FD_ADAM_SPLIT=0
done
