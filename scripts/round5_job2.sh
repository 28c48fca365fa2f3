#!/bin/bash
# round 5, GPU call 2: bench with the other configurations in their own processes; eager-only vs the eager/graph trial; packed-fp32 rates
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round5_bench_b.json.log 2> $O/round5_bench_b.stderr.log; echo "bench rc=$?"
L=$O/round5_eager_vs_trial.log; : > $L
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no_other_configs --no_cpu_baseline --no_roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('auto (eager + graph trial)', d['config']['launch'], d['windows']['ms_per_step'], d['windows']['host_issue_ms_per_step'])" >> $L
  python bench.py --eager --steps 20 --warmup 5 --no_other_configs --no_cpu_baseline --no_roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('--eager (no capture)     ', d['config']['launch'], d['windows']['ms_per_step'], d['windows']['host_issue_ms_per_step'])" >> $L
done
python bench.py --graph --steps 20 --warmup 5 --no_other_configs --no_cpu_baseline --no_roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('--graph                  ', d['config']['launch'], d['windows']['ms_per_step'], d['windows']['host_issue_ms_per_step'])" >> $L
./scripts/ubench/valu_rate3 > $O/round5_valu_rate3.log 2>&1
cat $L $O/round5_valu_rate3.log; cut -c1-300 $O/round5_bench_b.json.log; grep other_configs $O/round5_bench_b.stderr.log | cut -c1-420
