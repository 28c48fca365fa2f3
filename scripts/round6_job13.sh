#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_replay.py -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do
  FD_REPLAY_TRAIN=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null
  FD_REPLAY_TRAIN=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null
done | tee $O/round6_replay_train_ab.log
for v in 0 1; do FD_REPLAY_TRAIN=$v timeout 300 python scripts/secondary_ab.py r50 5 16 2>/dev/null; done | tee -a $O/round6_replay_train_ab.log
FD_REPLAY_TRAIN=1 timeout 600 python scripts/host_profile.py 12 2>&1 | grep -v amdgpu | head -8
