#!/bin/bash
# final-build evidence (after k_conv_wino2d_limb): smoke, whole GPU suite, probe counters on the final conv_wino.hip, the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r6f_smoke.log 2>&1; tail -6 $O/r6f_smoke.log
timeout 3000 python -m pytest tests -q -m gpu > $O/r6f_final_tests.log 2>&1; tail -4 $O/r6f_final_tests.log
bash scripts/pmc_probe.sh > $O/pmc_probe_round6f.log 2>&1; tail -3 $O/pmc_probe_round6f.log
cp $O/pmc_probe_wino.json profiles/round6_pmc_probe_wino.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_f.json.log 2> $O/round6_bench_f.stderr.log ) 2>&1 | tail -3
cut -c1-200 $O/round6_bench_f.json.log; grep -c "left null" $O/round6_bench_f.stderr.log; true
