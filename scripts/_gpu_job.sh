cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout -k 10 900 python -m pytest tests -x -q -m gpu ) > gpurun_out/r3_tests37.log 2>&1; grep -n "passed\|failed\|FAILED" gpurun_out/r3_tests37.log | tail -4
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -4
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/final_bench.err | head -2; python -c "
import json; d=json.loads(open('gpurun_out/final_bench.json').read().splitlines()[-1]); print(round(d['value'],1), round(d['ms_per_step'],3), d['final_loss'], round(d['step_mfma_frac'],3), round(d['roofline']['frac'],3), d['roofline']['traffic'], round(d['cpu_baseline']['value'],2))"
