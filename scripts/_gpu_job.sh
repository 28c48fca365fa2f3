cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_convstack.py -q -x -k "winograd or conv2d_fwd_bwd or decoder or encoder" 2>&1 | tail -3
python scripts/wgrad_probe.py 12 2>&1 | grep layer | cut -c1-80
bash scripts/pmc_kernel.sh r3_wgx k_wgrad_wino 2 -- python $R/scripts/wgrad_one.py 64 64 48 160 12 8 2>/dev/null | grep "us per launch\|SQ_INSTS_VALU \|SQ_INSTS_SALU\|SQ_INSTS_LDS\|VALU : MFMA\|matrix pipe\|BANK_CONFLICT"
run() { echo -n "$1 [$2] : "; ( cd $1 && env $2 python bench.py --steps 20 --warmup 5 --no_roofline --no_cpu_baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | tr '\n' ' ' ); echo; }
for i in 1 2 3; do
  run _ab_old "A=1"
  run . "A=1"
done
