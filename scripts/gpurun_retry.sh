#!/bin/bash
# usage: scripts/gpurun_retry.sh <timeout-seconds> '<command>'   - retries while the pool has no free slot (exit code 3)
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
