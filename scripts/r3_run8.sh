cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_convstack.py -q -x 2>&1 | tail -3) > gpurun_out/r3_conv_tests.log 2>&1; cat gpurun_out/r3_conv_tests.log
L=gpurun_out/r3_ksweep_ablate3.log; : > $L
for v in "" woob wnolds; do
  echo "=== variant '$v'" >> $L
  if [ -n "$v" ]; then export FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_$v.so; else unset FD_LIBFDHIP; fi
  timeout 200 python scripts/wino_ksweep.py 8 2>&1 | grep -v amdgpu.ids | tail -5 >> $L
  timeout 200 python scripts/wino_ksweep.py 24 2>&1 | grep -v amdgpu.ids | tail -5 >> $L
done
unset FD_LIBFDHIP
cat $L
timeout 600 python bench.py --no_cpu_baseline --steps 20 > gpurun_out/r3_bench5.json 2> gpurun_out/r3_bench5.err; echo bench rc $?; tail -2 gpurun_out/r3_bench5.err; python -c "
import json,sys; r=json.loads(open('gpurun_out/r3_bench5.json').read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['final_loss'], r['roofline']['frac'], r['roofline']['us_per_launch'], r['roofline_loss_path']['frac'])"
