#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for v in 0 1; do
  echo "== FD_LIMB_1X1=$v"
  FD_LIMB_1X1=$v timeout 1500 python -m pytest tests/test_gpu_trainer.py -q -m gpu -k "stacked_backward_against_float64 or 1024x320_batch8_backward" 2>&1 | grep -E "backward: (depth|pose|pose_encoder) |passed|failed" | cut -c1-200
done | tee $O/round6_backward_tests_limb_ab.log
timeout 900 python -m pytest tests/test_gpu_replay.py -x -q -m gpu 2>&1 | tail -15
