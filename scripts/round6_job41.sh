#!/bin/bash
# k_conv_wino2d_limb: parity, per-shape time, step A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "slab_kernel_split" 2>&1 | tail -20
cd scripts; timeout 300 python wino2d_limb_time.py 2>&1 | grep -v amdgpu; cd ..
for i in 1 2 3; do
  FD_WINO_FWD_LIMB=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_WINO_FWD_LIMB=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
