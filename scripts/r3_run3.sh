cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_convstack.py -q -x 2>&1 | tail -5) > gpurun_out/r3_conv_tests.log 2>&1; cat gpurun_out/r3_conv_tests.log
timeout 300 python scripts/wino_probe.py 12 > gpurun_out/r3_wino_probe.log 2>&1; cat gpurun_out/r3_wino_probe.log
timeout 300 python scripts/wino_probe.py 24 >> gpurun_out/r3_wino_probe.log 2>&1; tail -8 gpurun_out/r3_wino_probe.log
timeout 600 python bench.py --no_cpu_baseline --steps 20 > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.err; echo bench rc $?; tail -3 gpurun_out/r3_bench3.err; python -c "
import json; r=json.load(open('gpurun_out/r3_bench3.json')); print(r['value'], r['ms_per_step'], r['final_loss'], r['roofline']['frac'], r['roofline']['us_per_launch'], r['roofline_loss_path']['frac'])"
