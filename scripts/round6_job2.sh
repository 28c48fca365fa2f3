#!/bin/bash
# smoke + whole GPU suite + the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; tail -6 $O/r6_smoke.log
timeout 3000 python -m pytest tests -x -q -m gpu > $O/r6_job2_tests.log 2>&1; tail -25 $O/r6_job2_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_b.json.log 2> $O/round6_bench_b.stderr.log; tail -c 600 $O/round6_bench_b.json.log
