#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_refiner.py -x -q -m gpu 2>&1 | tail -12
for v in 1 0 1 0; do FD_PAD_ODD_CHANNELS=$v python bench.py --_other refiner_640x192 2>/dev/null | cut -c1-200; done
