#!/bin/bash
# in-step sweep of the limb GEMMs' split targets / depth on the ResNet-50 step (own process per variant, two rounds)
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2; do
  timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
  FD_LIMB_TARGET=1 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
  FD_LIMB_TARGET=128 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
  FD_LIMB_WGRAD_TARGET=128 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
  FD_LIMB_WGRAD_TARGET=512 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
  FD_LIMB_DEPTH=4 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null
done | tee $O/round6_limb_targets_step.log
