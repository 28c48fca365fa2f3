#!/bin/bash
# PMC passes of the roofline probe kernel on the round's final conv_wino.hip (bench.py checks the source's sha256), then the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
bash scripts/pmc_probe.sh > $O/pmc_probe_round6f.log 2>&1; tail -5 $O/pmc_probe_round6f.log
cp $O/pmc_probe_wino.json profiles/round6_pmc_probe_wino.json
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/round6_bench_g.json.log 2> $O/round6_bench_g.stderr.log ) 2>&1 | tail -3
cut -c1-200 $O/round6_bench_g.json.log; grep -c "left null" $O/round6_bench_g.stderr.log
