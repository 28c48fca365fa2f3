#!/bin/bash
# k_photo_ms duration vs batch (= workgroups in flight) and rows per strip, from rocprofv3 kernel traces   (run on the GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "3 0" "6 0" "12 0" "24 0" "12 40" "12 48" "12 64" "12 96" "12 16"; do
  set -- $cfg
  rm -rf /tmp/prof_occ
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_occ -- python -u $R/scripts/time_ms_kernel.py $1 $2 1 > /tmp/occ.log 2>&1
  DB=$(find /tmp/prof_occ -name "*_results.db" | head -1)
  python - "$DB" "$1" "$2" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
for n, k, avg, mn in db.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%k_photo_ms<%' group by 1"):
    print("B=%s rows=%s  %s  calls=%d avg=%.1f us min=%.1f us" % (sys.argv[2], sys.argv[3], n[-24:], k, avg, mn))
PY
done
