cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 2000 python -m pytest tests -q -m gpu -x --durations=5 ) > gpurun_out/r3_tests10.log 2>&1
grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_tests10.log | tail -5
python scripts/determinism_check.py 2>&1 | tail -6
