#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_limb.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python scripts/limb_ab.py 8 2>&1 | grep -v amdgpu | cut -c1-100,126-200,330-400 > $O/limb_ab_10.log; tail -14 $O/limb_ab_10.log
for i in 1 2 3; do
  FD_LIMB_KGROUPS=1 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
  FD_LIMB_TARGET=128 timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null
done | tee $O/round6_limb_step_ab_kg.log
