#!/bin/bash
# round 5: full GPU suite + smoke + the driver's bench command on the current commit
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1800 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -60 > $O/round5_gpu_tests_tail.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_bench_c.json.log 2> $O/round5_bench_c.stderr.log ) 2>&1 | tail -4
tail -5 $O/round5_gpu_tests_tail.log; grep -i "absrel" $O/round5_gpu_tests_tail.log | cut -c1-260; cut -c1-260 $O/round5_bench_c.json.log; grep other_configs $O/round5_bench_c.stderr.log | cut -c1-200
