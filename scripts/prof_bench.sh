#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py (eager path) -> gpurun_out/<tag>_bench_kernel_stats.md   (run on the GPU box)
TAG=${1:-r2}; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profb_$TAG
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/profb_$TAG -- python -u $R/bench.py --eager --steps 4 --warmup 2 --no_cpu_baseline --no_roofline "$@" > $R/gpurun_out/profb_$TAG.log 2>&1
echo "rocprof rc=$?"; tail -2 $R/gpurun_out/profb_$TAG.log | cut -c1-300
DB=$(find /tmp/profb_$TAG -name "*_results.db" | head -1)
# 2 steady-state + 2 warm-up + 4 timed = 8 optimiser steps traced
python $R/scripts/rocprof_summary.py $DB 8 50 k_adam_dev 5 > $R/gpurun_out/${TAG}_bench_kernel_stats.md
head -60 $R/gpurun_out/${TAG}_bench_kernel_stats.md
python $R/scripts/rocprof_timeline.py $DB > $R/gpurun_out/${TAG}_bench_timeline.md
cat $R/gpurun_out/${TAG}_bench_timeline.md
python $R/scripts/rocprof_top.py $DB 7 > $R/gpurun_out/${TAG}_bench_top_dispatches.md
python $R/scripts/rocprof_stream_chain.py $DB k_adam_dev k_photo_ms > $R/gpurun_out/${TAG}_bench_stream_chain_main.md 2>&1   # the depth network's stream
python $R/scripts/rocprof_stream_chain.py $DB k_adam_dev 0 > $R/gpurun_out/${TAG}_bench_stream_chain_0.md 2>&1
