#!/bin/bash
# Everything profiles/round5_* of the final build is made of, in one GPU call:  bash scripts/round5_profiles.sh <suffix>   (run on the GPU box)
SUF=${1:-a}
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; O=gpurun_out
# 0. the PMC passes bench.py reads its `traffic` figures from (keyed by the kernel source's sha256), into profiles/ of this copy first
bash scripts/pmc_probe.sh > $O/round5_pmc_probe_raw.txt 2>&1; cp $O/pmc_probe_wino.json $O/round5_pmc_probe_wino.json
cp $O/round5_pmc_probe_wino.json profiles/round5_pmc_probe_wino.json
bash scripts/pmc_loss_ms.sh round5_loss 12 > /dev/null 2>&1; cp $O/pmc_loss.json $O/round5_pmc_loss.json; cp $O/round5_loss_pmc.txt $O/round5_pmc_loss_raw.txt
cp $O/round5_pmc_loss.json profiles/round5_pmc_loss.json
# 1. the driver's command, un-profiled (carries other_configs and both CPU baselines)
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_bench_$SUF.json.log 2> $O/round5_bench_$SUF.stderr.log; echo "bench rc=$?"
tail -6 $O/round5_bench_$SUF.stderr.log | cut -c1-300
# 2. kernel-trace summaries of the step and of the probes
bash scripts/prof_bench.sh round5$SUF --no_other_configs > /dev/null 2>&1
bash scripts/prof_probe.sh round5$SUF > /dev/null 2>&1
bash scripts/prof_bench.sh round5${SUF}_r50 --no_other_configs --num_layers 50 --batch_size 8 > /dev/null 2>&1
# 3. PMC passes of the round's new kernels: the fused stem tail (batch 24 without features[0], batch 12 with)
bash scripts/pmc_kernel.sh round5_stem_tail_b24 k_bn_relu_pool 6 -- python $R/scripts/stem_tail_one.py 24 8 0 > /dev/null 2>&1
bash scripts/pmc_kernel.sh round5_stem_tail_b12_feat k_bn_relu_pool 6 -- python $R/scripts/stem_tail_one.py 12 8 1 > /dev/null 2>&1
ls -la $O | grep round5
