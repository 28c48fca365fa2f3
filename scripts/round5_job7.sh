#!/bin/bash
# round 5: fused stem tail - unit tests, the trainer tests that would notice a wrong gradient, same-box A/B on three configurations
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_convstack.py tests/test_gpu_evaluate.py -x -q -m gpu -k "stem_tail or evaluate or resize or resnet_encoder" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_trainer.py -x -q -m gpu -k "full_step or stacked or trajectory or matches_oracle or other_baseline" 2>&1 | tail -6
L=$O/round5_stem_tail_ab.log; : > $L
for i in 1 2 3; do
  for v in 1 0; do FD_FUSED_STEM_TAIL=$v python scripts/secondary_ab.py r18 3 20 >> $L 2>/dev/null; done
done
for cfg in r50 r18big; do for v in 1 0 1 0; do FD_FUSED_STEM_TAIL=$v python scripts/secondary_ab.py $cfg >> $L 2>/dev/null; done; done
cut -c1-60 $L | paste - <(sed 's/.*median/median/' $L)
