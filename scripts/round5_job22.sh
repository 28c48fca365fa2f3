#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -60 > $O/round5_gpu_tests_tail.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
tail -3 $O/round5_gpu_tests_tail.log
bash scripts/round5_profiles.sh k 2>&1 | tail -8
