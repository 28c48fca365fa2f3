set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(time timeout 1500 python -m pytest tests -q -m gpu -x --durations=25 --deselect tests/test_gpu_trainer.py::test_absrel_after_equal_steps_vs_oracle_fixture) > gpurun_out/r3_tests1.log 2>&1
tail -60 gpurun_out/r3_tests1.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke1.log 2>&1; tail -3 gpurun_out/r3_smoke1.log
timeout 600 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err; echo bench rc $?; tail -5 gpurun_out/r3_bench1.err; cat gpurun_out/r3_bench1.json
timeout 120 scripts/ubench/mfma_ablate > gpurun_out/r3_ablate0.log 2>&1; cat gpurun_out/r3_ablate0.log
