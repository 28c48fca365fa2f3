#!/bin/bash
# counters of the Winograd weight gradient on ResNet layer3's shape (256 -> 256 @12x40, batch 24): k_wgrad_wino_limb and the f32 kernel
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
FD_WINO_WGRAD_LIMB=2 scripts/pmc_kernel.sh round6f_wgrad_wino_limb k_wgrad_wino 4 -- python $R/scripts/wgrad_one.py 256 256 12 40 24 8 > /dev/null
FD_WINO_WGRAD_LIMB=0 scripts/pmc_kernel.sh round6f_wgrad_wino_f32 k_wgrad_wino 4 -- python $R/scripts/wgrad_one.py 256 256 12 40 24 8 > /dev/null
grep -E "^## |MFMA|us per launch|VALU :|HBM traffic|per wave" gpurun_out/round6f_wgrad_wino_limb_pmc.md gpurun_out/round6f_wgrad_wino_f32_pmc.md | cut -c1-220
