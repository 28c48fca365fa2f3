#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3; do
  FD_LIMB_1X1=0 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  FD_LIMB_DEPTH=4 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
done | tee $O/round6_limb_step_ab_v6.log
