#!/bin/bash
# rocprofv3 kernel stats + timeline of the Refiner step (bench.py --_other refiner_640x192), steady state
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profr
R=$GRAFT_REPO_ROOT
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/profr -- python -u $R/bench.py --_other refiner_640x192 > $R/gpurun_out/profr.log 2>&1
DB=$(find /tmp/profr -name "*_results.db" | head -1)
python $R/scripts/rocprof_summary.py $DB 8 45 k_adam_dev 6 > $R/gpurun_out/round6_refiner_kernel_stats.md
python $R/scripts/rocprof_timeline.py $DB > $R/gpurun_out/round6_refiner_timeline.md
head -56 $R/gpurun_out/round6_refiner_kernel_stats.md; head -30 $R/gpurun_out/round6_refiner_timeline.md
