#!/bin/bash
# narrow wgrad with two row groups: parity tests, then Refiner step old build (_ab_old) vs new on the same box
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do
  (cd $R/_ab_old && timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-300)
  timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-300
done
