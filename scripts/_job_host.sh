#!/bin/bash
# host-side cost of the step: profile + the busy-wait experiment (same box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python scripts/host_profile.py 60 > gpurun_out/r4_host_profile.log 2>&1
for d in 0 1 2 4 0; do
  echo "== FD_HOST_DELAY_US=$d" >> gpurun_out/r4_host_delay.log
  FD_HOST_DELAY_US=$d python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_other_configs --no_roofline 2>&1 | grep -E "^\{|timed" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['ms_per_step'], r['windows']['ms_per_step'], r['windows']['host_issue_ms_per_step'])
    else: print(l.strip())
" >> gpurun_out/r4_host_delay.log
done
