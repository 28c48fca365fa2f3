cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 scripts/ubench/mfma_ablate2 > gpurun_out/r3_ablate2.log 2>&1; cat gpurun_out/r3_ablate2.log
timeout 600 python bench.py --no_cpu_baseline --no_roofline --steps 30 > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; echo bench rc $?; tail -3 gpurun_out/r3_bench2.err; cat gpurun_out/r3_bench2.json
