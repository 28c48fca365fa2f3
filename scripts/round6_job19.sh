#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_refiner.py -q -m gpu -x -k "prefetched or reproducible or golden" 2>&1 | tail -5
for i in 1 2; do
  FD_REFINER_PREFETCH=0 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
  FD_REFINER_PREFETCH=1 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-120
done
