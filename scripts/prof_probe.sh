#!/bin/bash
# rocprofv3 kernel-trace summary of bench.py's roofline probes alone (--probe_only) -> gpurun_out/<tag>_probe_kernel_stats.md:
# the per-kernel averages that bench.py's HIP-event timings of the same launches have to agree with.   (run on the GPU box)
TAG=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profp_$TAG
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/profp_$TAG -- python -u $R/bench.py --probe_only > $R/gpurun_out/profp_$TAG.log 2>&1
echo "rocprof rc=$?"; tail -1 $R/gpurun_out/profp_$TAG.log | cut -c1-700
DB=$(find /tmp/profp_$TAG -name "*_results.db" | head -1)
python $R/scripts/rocprof_summary.py $DB 1 30 > $R/gpurun_out/${TAG}_probe_kernel_stats.md
head -24 $R/gpurun_out/${TAG}_probe_kernel_stats.md
