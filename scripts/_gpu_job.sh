cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_trainer.py -q -x -k "side_stream" 2>&1 | tail -3
