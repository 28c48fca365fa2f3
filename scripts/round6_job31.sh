#!/bin/bash
cd $GRAFT_REPO_ROOT/scripts
timeout 600 python limb_s2_ab.py r50 8 16 2>&1 | tail -30
cp ../profiles/round6_limb_s2_ab_r50.log ../gpurun_out/
