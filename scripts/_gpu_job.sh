cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/r3_tests30.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests30.log | tail -4
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -4 gpurun_out/final_bench.err; cut -c1-260 gpurun_out/final_bench.json
( time python bench.py ) > gpurun_out/default_bench.json 2> gpurun_out/default_bench.err; echo "default bench rc=$?"; tail -3 gpurun_out/default_bench.err | head -2; cut -c1-200 gpurun_out/default_bench.json
python bench.py --gpus 2 --steps 2 --warmup 1 2>&1 | tail -1; echo "gpus2 rc=$?"
