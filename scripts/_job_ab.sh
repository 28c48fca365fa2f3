#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_stream_shift.log
: > $L
for rep in 1 2; do
for k in 0 1 2 3 4 5 6 7; do
  FD_STREAM_SHIFT=$k python scripts/step_time.py shift_$k >> $L 2>/dev/null
done
done
for q in 3 5 6; do GPU_MAX_HW_QUEUES=$q python scripts/step_time.py queues_$q >> $L 2>/dev/null; done
cat $L
