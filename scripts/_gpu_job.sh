cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "winograd" ) > gpurun_out/r3_t14.log 2>&1; grep -n "passed\|failed\|FAILED\|Error\|rel err\|assert" gpurun_out/r3_t14.log | tail -15
python scripts/conv_time.py 2>&1 | tail -9
for i in 1 2; do
for v in 1 0; do
echo "FD_WINO_FWD_2D=$v"; FD_WINO_FWD_2D=$v python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/bench_err_$v.txt | cut -c1-160; tail -2 gpurun_out/bench_err_$v.txt
done; done
