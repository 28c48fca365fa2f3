#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3 4; do
  timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  FD_LIMB_TARGET=1 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  FD_LIMB_TARGET=1 FD_LIMB_WGRAD_TARGET=128 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  FD_LIMB_1X1=0 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
done | tee $O/round6_limb_targets_step_b.log
