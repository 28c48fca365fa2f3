cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_trainer.py -q -m gpu -x -k "late_weight or side_stream or equal_steps or graph" ) > gpurun_out/r3_t17.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t17.log | tail -8
for i in 1 2; do
for v in 1 0; do
echo "FD_LATE_RELAYOUT=$v"; FD_LATE_RELAYOUT=$v python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/bench_err_$v.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['param_checksum'], d['final_loss'])"; tail -1 gpurun_out/bench_err_$v.txt
done; done
