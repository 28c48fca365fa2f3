#!/bin/bash
# stride-2 convolutions on the limb implicit GEMM: parity, per-shape A/B, step A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_limb.py -q -m gpu -x -k "stride2" 2>&1 | tail -25
cd scripts && timeout 600 python limb_s2_ab.py 12 24 2>&1 | tail -20; cd ..
cp profiles/round6_limb_s2_ab.log gpurun_out/
for i in 1 2; do
  FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
