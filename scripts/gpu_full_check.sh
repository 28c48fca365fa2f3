#!/bin/bash
# pytest -m gpu + smoke() + the driver's bench command in one GPU call (gpurun -- bash scripts/gpu_full_check.sh)
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4_full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round4_bench_c.json.log 2> gpurun_out/round4_bench_c.err ) 2> gpurun_out/r4_bench_walltime.log
cat gpurun_out/r4_full_gpu_tests.log; tail -2 gpurun_out/r4_smoke.log; cat gpurun_out/r4_bench_walltime.log; cut -c1-400 gpurun_out/round4_bench_c.json.log
