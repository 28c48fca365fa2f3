cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/round3_bench_f.json.log 2> gpurun_out/round3_bench_f.err; echo "bench rc=$?"; tail -3 gpurun_out/round3_bench_f.err
