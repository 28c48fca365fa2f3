#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3; do
  (cd _ab_old && timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null | sed 's/^/old /')
  FD_LIMB_1X1=0 timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null | sed 's/^/new /'
  FD_LIMB_1X1=1 timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null | sed 's/^/new /'
done | tee $O/round6_old_vs_new_r18_b.log
