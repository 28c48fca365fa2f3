#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2; do
  FD_SIDE_WGRAD=depth timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
  timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
done
timeout 1200 python -m pytest tests/test_gpu_refiner.py tests/test_gpu_replay.py -q -m gpu -x 2>&1 | tail -5
