#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r4_full_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4_smoke.log 2>&1
cat gpurun_out/r4_full_gpu_tests.log; tail -2 gpurun_out/r4_smoke.log
