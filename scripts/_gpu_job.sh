cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scripts/host_profile.py 12 2>&1 | grep "un-profiled"
( timeout -k 10 600 python -m pytest tests/test_gpu_convstack.py -q -m gpu -k "winograd or n16 or reflect" ) > gpurun_out/r3_t21.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t21.log | tail -5
timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>&1 | grep "timed\|value" | cut -c1-200
