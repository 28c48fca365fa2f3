#!/bin/bash
# coalesced loader of the 1x1 limb weight gradient: parity, per-shape A/B, ResNet-50 step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_limb.py -q -m gpu -x 2>&1 | tail -3
cd scripts && timeout 600 python limb_wgrad_coal_ab.py 2>&1 | tail -16; cd ..
cp profiles/round6_limb_wgrad_coal_ab.log gpurun_out/
for i in 1 2; do
  FD_LIMB_WGRAD_COAL=0 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
  FD_LIMB_WGRAD_COAL=1 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
done
