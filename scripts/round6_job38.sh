#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -m gpu -x -k "round6_tuning or round5_tuning" 2>&1 | tail -3
