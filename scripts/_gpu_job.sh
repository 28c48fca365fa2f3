cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests -q -m gpu ) > gpurun_out/r3_tests24.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests24.log | tail -8; grep -n "AbsRel after" gpurun_out/r3_tests24.log | head -4
