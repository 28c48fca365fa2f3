#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_convstack.py tests/test_gpu_trainer.py -x -q -m gpu -k "not absrel" 2>&1 | tail -4
L=$O/round5_bn_remask_ab.log; : > $L
for i in 1 2 3; do for v in 1 0; do FD_BN_REMASK=$v python scripts/secondary_ab.py r18 3 20 >> $L 2>/dev/null; done; done
for i in 1 2; do for v in 1 0; do FD_BN_REMASK=$v python scripts/secondary_ab.py r50 >> $L 2>/dev/null; done; done
cut -c1-50 $L | paste - <(sed 's/.*median/median/' $L)
