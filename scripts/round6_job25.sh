#!/bin/bash
# limb stride-2 (3x3 only): tests, what still runs on the direct kernels, A/B of the other configurations
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_limb.py tests/test_gpu_convstack.py -q -m gpu -x 2>&1 | tail -4
timeout 300 python scripts/step_conv_log.py r18 2>&1 | python scripts/conv_log_summary.py > gpurun_out/round6_conv_routes_r18.log; head -40 gpurun_out/round6_conv_routes_r18.log
for i in 1 2; do
  FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
  FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
done
FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r18big 3 10 2>/dev/null | tail -1 | cut -c1-200
FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r18big 3 10 2>/dev/null | tail -1 | cut -c1-200
FD_LIMB_CONV=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
FD_LIMB_CONV=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
