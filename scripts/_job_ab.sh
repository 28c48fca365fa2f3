#!/bin/bash
# A/B of the current library against libfdhip_prev.so (previous commit's conv_wino.hip) on one box
cd $GRAFT_REPO_ROOT
L=gpurun_out/r4_ab_b64.log
: > $L
timeout 900 python -m pytest tests/test_gpu_convstack.py -x -q 2>&1 | tail -3 >> $L
echo "== conv2p_time (new)" >> $L; python scripts/conv2p_time.py >> $L 2>&1
echo "== conv2p_time (prev)" >> $L; FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_prev.so python scripts/conv2p_time.py >> $L 2>&1
for i in 1 2; do
  python scripts/step_time.py new >> $L 2>/dev/null
  FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_prev.so python scripts/step_time.py prev >> $L 2>/dev/null
done
cat $L
