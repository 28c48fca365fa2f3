#!/bin/bash
# rocprofv3 kernel-trace summary of the multi-scale loss probe -> gpurun_out/<tag>_kernel_stats.md   (run on the GPU box)
TAG=${1:-ms}; B=${2:-12}; ROWS=${3:-0}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_$TAG
PROBE_TIMING=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -- python -u $R/scripts/probe_loss_ms.py $B $ROWS 20 > $R/gpurun_out/prof_$TAG.log 2>&1
echo "rocprof rc=$?"
DB=$(find /tmp/prof_$TAG -name "*_results.db" | head -1)
python $R/scripts/rocprof_summary.py $DB 1 20 > $R/gpurun_out/${TAG}_kernel_stats.md
cat $R/gpurun_out/${TAG}_kernel_stats.md
