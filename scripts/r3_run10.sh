cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -q -m gpu -x --durations=12 --deselect tests/test_gpu_trainer.py::test_absrel_after_equal_steps_vs_oracle_fixture) > gpurun_out/r3_tests2.log 2>&1
grep -v "Warning\|warn\|^$\|run_backward\|np.array" gpurun_out/r3_tests2.log | tail -75
timeout 300 python scripts/phase_timing.py 5 2>&1 | grep -v amdgpu.ids > gpurun_out/r3_phase.log; cat gpurun_out/r3_phase.log
for ns in 1 2 4; do echo "FD_NSTREAMS=$ns"; FD_NSTREAMS=$ns timeout 300 python bench.py --no_cpu_baseline --no_roofline --eager --steps 15 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'])"; done
