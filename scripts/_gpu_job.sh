cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -k "winograd_weight_gradient or wgrad or conv" ) > gpurun_out/r3_t13.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t13.log | tail -15
for i in 1 2; do
for v in 1 0; do
echo "FD_WINO_WGRAD_2D=$v"; FD_WINO_WGRAD_2D=$v python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/bench_err_$v.txt | cut -c1-160; tail -3 gpurun_out/bench_err_$v.txt
done; done
