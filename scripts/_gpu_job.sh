cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests -q -m gpu ) > gpurun_out/r3_tests27.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests27.log | tail -8; grep -n "AbsRel after" gpurun_out/r3_tests27.log | head -4
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
timeout -k 10 900 bash scripts/round3_profiles.sh g 2>&1 | grep -v "^-rw" | tail -12
