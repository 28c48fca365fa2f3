cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 600 python -m pytest tests/test_gpu_convstack.py -q -m gpu -k "n16" ) > gpurun_out/r3_t18.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t18.log | tail -12
timeout 120 python scripts/n16_time.py 2>&1 | tail -5
for i in 1 2; do
for v in 1 0; do
echo "FD_CONV_N16=$v"; FD_CONV_N16=$v timeout 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/bench_err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"; tail -1 gpurun_out/bench_err_$v.txt
done; done
