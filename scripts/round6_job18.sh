#!/bin/bash
# 2-D Winograd weight gradient for Cin % 32 == 16: parity, then Refiner A/B (wino_wgrad_2d = 1 vs 2) and the headline configs
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "winograd_weight_gradient" 2>&1 | tail -3
for i in 1 2; do
  FD_WINO_WGRAD_2D=1 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
  FD_WINO_WGRAD_2D=2 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
done
