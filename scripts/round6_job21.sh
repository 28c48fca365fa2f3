#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_refiner.py tests/test_gpu_replay.py -q -m gpu -x 2>&1 | tail -5
for i in 1 2; do
  FD_FOLD_FROZEN_BN=0 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
  timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
done
