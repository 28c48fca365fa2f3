#!/bin/bash
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
for i in 1 2 3; do
  for v in 0 1; do
    FD_REPLAY_FROZEN=$v timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('refiner FD_REPLAY_FROZEN=%s  %.1f images/s  %.2f ms  windows %s' % (os.environ.get('FD_REPLAY_FROZEN'), d['value'], d['ms_per_step'], ['%.2f'%w for w in d['windows_ms_per_step']]))"
  done
done | tee $O/round6_refiner_replay_ab.log
