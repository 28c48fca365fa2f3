#!/bin/bash
# PMC passes over the limb GEMM kernel on two ResNet-50 bottleneck shapes (an HBM-heavy shallow one, a deep one)
R=$GRAFT_REPO_ROOT
scripts/pmc_kernel.sh limb_l1 k_gemm_limb 4 -- python $R/scripts/conv_one.py 256 48 160 128 1 1 0 8 12 > /dev/null
scripts/pmc_kernel.sh limb_l3 k_gemm_limb 4 -- python $R/scripts/conv_one.py 1024 12 40 512 1 1 0 8 12 > /dev/null
cat $R/gpurun_out/limb_l1_pmc.md $R/gpurun_out/limb_l3_pmc.md
