#!/bin/bash
# k_wgrad_wino_limb<REFL>: parity, Refiner / headline / 1024x320 A/B of wino_wgrad_limb 1 vs 2
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "split_precision" 2>&1 | tail -3
for i in 1 2 3; do
  FD_WINO_WGRAD_LIMB=1 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-60
  FD_WINO_WGRAD_LIMB=2 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-60
done
for i in 1 2; do
  FD_WINO_WGRAD_LIMB=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_WINO_WGRAD_LIMB=2 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
