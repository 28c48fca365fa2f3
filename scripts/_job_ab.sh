#!/bin/bash
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_ab_m128b.log
: > $L
timeout 900 python -m pytest tests/test_gpu_convstack.py -x -q 2>&1 | tail -2 >> $L
python scripts/conv2d_m128_time.py 2>&1 | grep batch >> $L
python scripts/conv2d_m128_time.py 2>&1 | grep "256 -> 128" >> $L
for i in 1 2; do
  python scripts/step_time.py m128_rule >> $L 2>/dev/null
  FD_WINO_FWD_2D_M128=0 python scripts/step_time.py m128_off >> $L 2>/dev/null
done
cat $L
