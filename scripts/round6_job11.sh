#!/bin/bash
# final-build evidence: smoke, whole GPU suite, the driver's bench command, rocprofv3 kernel stats (headline + ResNet-50)
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; tail -5 $O/r6_smoke.log
timeout 3000 python -m pytest tests -q -m gpu > $O/r6_final_tests.log 2>&1; tail -4 $O/r6_final_tests.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_c.json.log 2> $O/round6_bench_c.stderr.log; tail -c 300 $O/round6_bench_c.json.log
scripts/prof_bench.sh round6b --no_other_configs > $O/prof_round6b.log 2>&1; head -12 $O/round6b_bench_kernel_stats.md
scripts/prof_bench.sh round6b_r50 --num_layers 50 --batch_size 8 --no_other_configs > $O/prof_round6b_r50.log 2>&1; head -12 $O/round6b_r50_bench_kernel_stats.md
