#!/bin/bash
# round 5, first GPU call: GPU suite on the round-start code (+ ADVICE fixes), the driver's bench command, secondary-config A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -8 > $O/r5_job1_tests.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_bench_a.json.log 2> $O/round5_bench_a.stderr.log; echo "bench rc=$?"
L=$O/round5_secondary_ab.log; : > $L
for cfg in r50 r18big; do
  python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_WINO_FWD_2D_M128=0 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_WINO_WGRAD_TARGET=384 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_WINO_FWD_2DP_MIN=0 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_WINO_TARGET=256 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_FUSED_CONV_BN=0 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  FD_DECODER_FUSED_ACT=0 python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
  python scripts/secondary_ab.py $cfg >> $L 2>/dev/null
done
cat $O/r5_job1_tests.log; cat $L; cut -c1-600 $O/round5_bench_a.json.log; tail -12 $O/round5_bench_a.stderr.log
