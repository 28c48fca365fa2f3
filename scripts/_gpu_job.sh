cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/r3_tests34.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests34.log | tail -4
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/final_bench.err | head -2; cut -c1-200 gpurun_out/final_bench.json
