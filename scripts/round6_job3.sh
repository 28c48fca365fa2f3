#!/bin/bash
# same-box A/B: round-5 tree (_ab_old, commit d6f0709) vs this tree, headline + refiner-sized configs, alternating
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3; do
  (cd _ab_old && timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null | sed 's/^/old /')
  timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null | sed 's/^/new /'
done | tee $O/round6_old_vs_new_r18.log
timeout 900 python -m pytest tests/test_gpu_refiner.py tests/test_gpu_completor.py -x -q -m gpu 2>&1 | tail -12
