cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --eager --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3))"; grep timed gpurun_out/err.txt; }
run FD_STAGED_BWD=1
run FD_STAGED_BWD=0
run "FD_STAGED_BWD=1 FD_BWD_ORDER=depth,beam,pose,beampose"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pst && FD_STAGED_BWD=1 timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/pst -- python -u $GRAFT_REPO_ROOT/bench.py --eager --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > /dev/null 2>&1
DB=$(find /tmp/pst -name "*_results.db" | head -1); python $GRAFT_REPO_ROOT/scripts/rocprof_timeline.py $DB | tail -12
rm -rf /tmp/pst0 && FD_STAGED_BWD=0 timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/pst0 -- python -u $GRAFT_REPO_ROOT/bench.py --eager --steps 4 --warmup 2 --no_cpu_baseline --no_roofline > /dev/null 2>&1
DB=$(find /tmp/pst0 -name "*_results.db" | head -1); python $GRAFT_REPO_ROOT/scripts/rocprof_timeline.py $DB | tail -12
