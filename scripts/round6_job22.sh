#!/bin/bash
# full check of the tree: pytest -m gpu, smoke(), the driver's bench command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -q -m gpu 2>&1 | tail -40 > gpurun_out/r6e_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r6e_smoke.log 2>&1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round6_bench_e.json.log 2> gpurun_out/round6_bench_e.err ) 2> gpurun_out/r6e_bench_walltime.log
tail -8 gpurun_out/r6e_gpu_tests.log; tail -2 gpurun_out/r6e_smoke.log; cat gpurun_out/r6e_bench_walltime.log; cut -c1-600 gpurun_out/round6_bench_e.json.log
