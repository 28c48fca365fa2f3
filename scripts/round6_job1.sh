#!/bin/bash
# limb tests + conv stack tests + R50 / R18 step A/B (limb on / off, alternating, own processes)
cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_limb.py tests/test_gpu_convstack.py -x -q -m gpu > $O/r6_job1_tests.log 2>&1; tail -15 $O/r6_job1_tests.log
for i in 1 2 3; do
  for v in 0 1; do FD_LIMB_1X1=$v timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null; done
done | tee $O/round6_limb_step_ab.log
for v in 0 1; do FD_LIMB_1X1=$v timeout 300 python scripts/secondary_ab.py r18 3 20 2>/dev/null; done | tee -a $O/round6_limb_step_ab.log
