cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -6
( time python bench.py ) > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default bench rc=$?"; tail -3 gpurun_out/bench_default.err; cut -c1-400 gpurun_out/bench_default.json
python bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -2; echo "gpus2 rc=$?"
( time timeout 2000 python -m pytest tests -q -m gpu -x ) > gpurun_out/r3_tests11.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests11.log | tail -3; tail -4 gpurun_out/r3_tests11.log | head -3
