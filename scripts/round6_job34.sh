#!/bin/bash
# final-build evidence 2: rocprofv3 kernel stats (headline, ResNet-50, Refiner, roofline probes), counters of the limb stride-2 kernels
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
scripts/prof_bench.sh round6f --no_other_configs > $O/prof_round6f.log 2>&1; head -30 $O/round6f_bench_kernel_stats.md
scripts/prof_bench.sh round6f_r50 --num_layers 50 --batch_size 8 --no_other_configs > $O/prof_round6f_r50.log 2>&1; head -20 $O/round6f_r50_bench_kernel_stats.md
scripts/prof_probe.sh round6f > $O/prof_round6f_probe.log 2>&1; head -16 $O/round6f_probe_kernel_stats.md
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profr && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/profr -- python -u $R/bench.py --_other refiner_640x192 > $R/gpurun_out/profr.log 2>&1
  DB=$(find /tmp/profr -name "*_results.db" | head -1)
  python $R/scripts/rocprof_summary.py $DB 8 45 k_adam_dev 6 > $R/gpurun_out/round6f_refiner_kernel_stats.md )
head -20 $O/round6f_refiner_kernel_stats.md
scripts/pmc_kernel.sh round6f_limb_s2 k_conv_limb 4 -- python $R/scripts/conv_one.py 64 48 160 128 3 2 1 24 12 > /dev/null
cat $O/round6f_limb_s2_pmc.md | head -80
