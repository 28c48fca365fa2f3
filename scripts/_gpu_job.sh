cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "winograd" ) > gpurun_out/r3_t16.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t16.log | tail -5
python scripts/conv_time.py 2>&1 | grep "256 -> 256\|512 -> 512"
FD_WINO_2D_MAP=0 python scripts/conv_time.py 2>&1 | grep "256 -> 256\|512 -> 512"
for i in 1 2; do
for v in 1 0; do
echo "FD_WINO_2D_MAP=$v"; FD_WINO_2D_MAP=$v python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/bench_err_$v.txt | cut -c1-160; tail -1 gpurun_out/bench_err_$v.txt
done; done
bash scripts/pmc_kernel.sh round3_conv_wino2d_l4 k_conv_wino2d 2 -- python $GRAFT_REPO_ROOT/scripts/conv_one.py 512 6 20 512 3 1 1 24 8 > /dev/null 2>&1; tail -6 gpurun_out/round3_conv_wino2d_l4_pmc.md
