#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_convstack.py -x -q 2>&1 | tail -2
python scripts/stem_time.py 2>&1 | grep gradient
for i in 1 2; do python scripts/step_time.py new 2>/dev/null; done
