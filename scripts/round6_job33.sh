#!/bin/bash
# final-build evidence 1: smoke, whole GPU suite, the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6f_smoke.log 2>&1; tail -6 $O/r6f_smoke.log
timeout 3000 python -m pytest tests -q -m gpu > $O/r6f_final_tests.log 2>&1; tail -4 $O/r6f_final_tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_f.json.log 2> $O/round6_bench_f.stderr.log ) 2>&1 | tail -3
cut -c1-260 $O/round6_bench_f.json.log
