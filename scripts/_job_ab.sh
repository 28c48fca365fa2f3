#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_convbn2.log
: > $L
for i in 1 2 3 4 5; do
python scripts/step_time.py fused >> $L 2>/dev/null
FD_FUSED_CONV_BN=0 python scripts/step_time.py unfused >> $L 2>/dev/null
done
cat $L
