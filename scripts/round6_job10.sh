#!/bin/bash
# same-box A/B of the headline configuration: round-5 tree (_ab_old) vs this tree, 5 alternating pairs, 5 windows of 20 steps
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3 4 5; do
  (cd _ab_old && timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | sed 's/^/old /')
  timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | sed 's/^/new /'
done | tee $O/round6_old_vs_new_r18_c.log
