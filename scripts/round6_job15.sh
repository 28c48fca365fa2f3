#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out
scripts/prof_bench.sh round6c --no_other_configs > $O/prof_round6c.log 2>&1; head -64 $O/round6c_bench_kernel_stats.md
scripts/prof_bench.sh round6c_r50 --num_layers 50 --batch_size 8 --no_other_configs > $O/prof_round6c_r50.log 2>&1; head -14 $O/round6c_r50_bench_kernel_stats.md
