#!/bin/bash
# Everything profiles/round3_* is made of, in one GPU call:  bash scripts/round3_profiles.sh <suffix>   (run on the GPU box)
SUF=${1:-a}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out
# 0. the PMC passes bench.py reads its `traffic` figures from (keyed by the kernel source's sha256), into profiles/ of this copy first
bash scripts/pmc_probe.sh > $O/round3_pmc_probe_raw.txt 2>&1; cp $O/pmc_probe_wino.json $O/round3_pmc_probe_wino.json
cp $O/round3_pmc_probe_wino.json profiles/round3_pmc_probe_wino.json
bash scripts/pmc_loss_ms.sh round3_loss 12 > /dev/null 2>&1; cp $O/pmc_loss.json $O/round3_pmc_loss.json; cp $O/round3_loss_pmc.txt $O/round3_pmc_loss_raw.txt
cp $O/round3_pmc_loss.json profiles/round3_pmc_loss.json
# 1. bench lines (the driver's command first), un-profiled
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round3_bench_$SUF.json.log 2> $O/round3_bench_$SUF.err; echo "bench rc=$?"
tail -3 $O/round3_bench_$SUF.err
python bench.py --num_layers 50 --batch_size 8 --no_cpu_baseline > $O/round3_bench_${SUF}_resnet50_b8.json.log 2>/dev/null; echo "r50 rc=$?"
python bench.py --height 320 --width 1024 --batch_size 8 --no_cpu_baseline > $O/round3_bench_${SUF}_1024x320_b8.json.log 2>/dev/null; echo "1024 rc=$?"
# 2. kernel-trace summaries of the step and of the probes
bash scripts/prof_bench.sh round3$SUF > /dev/null 2>&1
bash scripts/prof_probe.sh round3$SUF > /dev/null 2>&1
# 3. PMC passes: probe kernel (-> json for bench.py), loss kernels (-> json + raw), Winograd weight gradient, direct kernel
bash scripts/pmc_kernel.sh round3_wgrad_wino_l1 k_wgrad_wino 2 -- python $R/scripts/wgrad_one.py 64 64 48 160 12 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round3_wgrad_wino_l3 k_wgrad_wino 2 -- python $R/scripts/wgrad_one.py 256 256 12 40 24 8 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round3_conv_wino_l1 k_conv_wino 2 -- python $R/scripts/probe_layer1.py 12 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round3_conv_fast_s2 k_conv_fast 4 -- python $R/scripts/conv_one.py 64 48 160 128 3 2 1 24 8 > /dev/null 2>&1
ls -la $O | grep round3
bash scripts/pmc_kernel.sh round3_conv_wino2d_l4 k_conv_wino2d 2 -- python $R/scripts/conv_one.py 512 6 20 512 3 1 1 24 8 > /dev/null 2>&1
ls -la $O | grep round3
