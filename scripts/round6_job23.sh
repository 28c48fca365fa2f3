#!/bin/bash
# early loss inputs (identity losses + noise beside the encoders): parity, then same-box A/B; RCCL one-rank communicator test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_rccl_single_rank.py tests/test_gpu_trainer.py -q -m gpu -x -k "rccl or early_loss or side_streams or reproducible" 2>&1 | tail -8
for i in 1 2 3; do
  FD_EARLY_LOSS_INPUTS=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_EARLY_LOSS_INPUTS=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
timeout 300 python bench.py --_other completor_1216x352 2>/dev/null | tail -1 | cut -c1-400
timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-400
