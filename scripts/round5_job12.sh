#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_convstack.py tests/test_gpu_trainer.py tests/test_gpu_refiner.py tests/test_gpu_completor.py tests/test_gpu_ablations.py -x -q -m gpu -k "not absrel" 2>&1 | tail -4
git stash -q 2>/dev/null
L=$O/round5_stack_norm_ab.log; : > $L
for i in 1 2 3 4; do python scripts/secondary_ab.py r18 3 20 >> $L 2>/dev/null; done
cut -c1-30 $L | paste - <(sed 's/.*median/median/' $L)
