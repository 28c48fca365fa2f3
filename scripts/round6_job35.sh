#!/bin/bash
# k_wgrad_wino_limb: parity, per-shape time against the f32 kernel, step A/B
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "split_precision" 2>&1 | tail -14
echo "f32:"; FD_WINO_WGRAD_LIMB=0 timeout 200 python scripts/wgrad_time.py
echo "limb:"; FD_WINO_WGRAD_LIMB=1 timeout 200 python scripts/wgrad_time.py
for i in 1 2; do
  FD_WINO_WGRAD_LIMB=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_WINO_WGRAD_LIMB=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
