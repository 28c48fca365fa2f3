#!/bin/bash
# large 1x1 stride-2 layers (ResNet-50 downsample) on the limb kernels: parity, ResNet-50 / ResNet-18 step A/B against limb_conv = 0
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_limb.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python scripts/step_conv_log.py r50 2>&1 | python scripts/conv_log_summary.py | grep -v "wino\|refl\|limb 1x1" | head -24
for i in 1 2; do
  FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
  FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
done
FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
