cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2000 python -m pytest tests -q -m gpu -x --durations=5 ) > gpurun_out/r3_tests8.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_tests8.log | tail -5
