#!/bin/bash
# final-build evidence: the driver's bench command, rocprofv3 kernel stats (headline + ResNet-50), smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6_smoke.log 2>&1; tail -4 $O/r6_smoke.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_d.json.log 2> $O/round6_bench_d.stderr.log; tail -c 300 $O/round6_bench_d.json.log
scripts/prof_bench.sh round6c --no_other_configs > $O/prof_round6c.log 2>&1; head -12 $O/round6c_bench_kernel_stats.md
scripts/prof_bench.sh round6c_r50 --num_layers 50 --batch_size 8 --no_other_configs > $O/prof_round6c_r50.log 2>&1; head -12 $O/round6c_r50_bench_kernel_stats.md
