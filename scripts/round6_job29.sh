#!/bin/bash
# which convolutions of the other configurations still run on the f32 direct kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in r50 completor r18big; do
  timeout 300 python scripts/step_conv_log.py $c 2>&1 | python scripts/conv_log_summary.py > gpurun_out/round6_conv_routes_$c.log
  echo "== $c"; grep -v "wino\|limb\|refl" gpurun_out/round6_conv_routes_$c.log | head -30
done
timeout 300 python scripts/refiner_conv_log.py 2>&1 | tail -30
