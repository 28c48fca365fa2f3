#!/bin/bash
# grouped limb launches with few workgroups on 64-row tiles: parity under the switch, step A/B
cd $GRAFT_REPO_ROOT
FD_LIMB_GRP_MB2=256 timeout 600 python -m pytest tests/test_gpu_limb.py -q -m gpu -x -k stride2 2>&1 | tail -2
for i in 1 2 3; do
  FD_LIMB_GRP_MB2=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_LIMB_GRP_MB2=256 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
FD_LIMB_GRP_MB2=0 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
FD_LIMB_GRP_MB2=256 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
