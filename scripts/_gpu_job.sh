cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_trainer.py -q -m gpu -k "side_streams or side_stream or late_weight" ) > gpurun_out/r3_t28.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t28.log | tail -8
