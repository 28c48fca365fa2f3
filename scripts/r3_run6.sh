cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_convstack.py -q -x 2>&1 | tail -5) > gpurun_out/r3_conv_tests.log 2>&1; cat gpurun_out/r3_conv_tests.log
timeout 200 python scripts/wino_ksweep.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_ksweep2.log
timeout 200 python scripts/wino_ksweep.py 12 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_ksweep2.log
timeout 200 python scripts/wino_ksweep.py 24 2>&1 | grep -v amdgpu.ids >> gpurun_out/r3_ksweep2.log
cat gpurun_out/r3_ksweep2.log
timeout 300 python scripts/wino_probe.py 12 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_wino_probe2.log; cat gpurun_out/r3_wino_probe2.log
timeout 600 python bench.py --no_cpu_baseline --steps 20 > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err; echo bench rc $?; tail -3 gpurun_out/r3_bench4.err; python -c "
import json,sys; r=json.loads(open('gpurun_out/r3_bench4.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['final_loss'], r['roofline']['frac'], r['roofline']['us_per_launch'], r['roofline_loss_path']['frac'])"
