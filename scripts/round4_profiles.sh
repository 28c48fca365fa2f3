#!/bin/bash
# Everything profiles/round4_* is made of, in one GPU call:  bash scripts/round4_profiles.sh <suffix>   (run on the GPU box)
SUF=${1:-a}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out
# 0. the PMC passes bench.py reads its `traffic` figures from (keyed by the kernel source's sha256), into profiles/ of this copy first
bash scripts/pmc_probe.sh > $O/round4_pmc_probe_raw.txt 2>&1; cp $O/pmc_probe_wino.json $O/round4_pmc_probe_wino.json
cp $O/round4_pmc_probe_wino.json profiles/round4_pmc_probe_wino.json
bash scripts/pmc_loss_ms.sh round4_loss 12 > /dev/null 2>&1; cp $O/pmc_loss.json $O/round4_pmc_loss.json; cp $O/round4_loss_pmc.txt $O/round4_pmc_loss_raw.txt
cp $O/round4_pmc_loss.json profiles/round4_pmc_loss.json
# 1. the driver's command, un-profiled (carries other_configs and both CPU baselines)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round4_bench_$SUF.json.log 2> $O/round4_bench_$SUF.err; echo "bench rc=$?"
tail -8 $O/round4_bench_$SUF.err
# 2. kernel-trace summaries of the step and of the probes
bash scripts/prof_bench.sh round4$SUF --no_other_configs > /dev/null 2>&1
bash scripts/prof_probe.sh round4$SUF > /dev/null 2>&1
# 3. PMC passes: the one-workgroup F(2x2, 3x3) kernel on layer1 (batch 12 / 24), the slab kernel on layer4, the Winograd weight gradient, a 1x1 shape
bash scripts/pmc_kernel.sh round4_conv_wino2p_l1_b12 k_conv_wino2p 2 -- python $R/scripts/probe_w2p.py 12 1 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_conv_wino2p_l1_b24 k_conv_wino2p 2 -- python $R/scripts/probe_w2p.py 24 1 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_conv_wino2d_l4 k_conv_wino2d 2 -- python $R/scripts/conv_one.py 512 6 20 512 3 1 1 24 8 > /dev/null 2>&1
FD_WINO_FWD_2D_M128=0 bash scripts/pmc_kernel.sh round4_conv_wino2d_l4_m64 k_conv_wino2d 2 -- python $R/scripts/conv_one.py 512 6 20 512 3 1 1 24 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_wgrad_wino_l3 k_wgrad_wino 2 -- python $R/scripts/wgrad_one.py 256 256 12 40 24 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_wgrad_wino_l1 k_wgrad_wino 2 -- python $R/scripts/wgrad_one.py 64 64 48 160 12 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_conv_stem k_conv7s2_stem 2 -- python $R/scripts/conv_one.py 6 192 640 64 7 2 3 24 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round4_conv_1x1_r50 k_conv_fast 2 -- python $R/scripts/conv_one.py 256 48 160 64 1 1 0 8 8 > /dev/null 2>&1
ls -la $O | grep round4
