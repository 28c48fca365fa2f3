cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout -k 10 600 python -m pytest tests/test_gpu_convstack.py -q -m gpu -k "single_output or conv2d_fwd_bwd" ) > gpurun_out/r3_t31.log 2>&1; grep -n "passed\|failed\|FAILED\|Error" gpurun_out/r3_t31.log | tail -5
python - <<'PY'
import torch, sys
sys.path.insert(0, '.')
from fusiondepth_amd import functional as FD
for c, h, w in [(16, 192, 640), (32, 96, 320), (64, 48, 160), (128, 24, 80)]:
    x = torch.randn(12, c, h, w, device="cuda"); wt = torch.randn(1, c, 3, 3, device="cuda") * 0.1; b = torch.zeros(1, device="cuda")
    f = lambda: FD.conv2d(x, wt, b, 1, 1, "reflect", "sigmoid")
    with torch.no_grad():
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
    print("dispconv %3d -> 1 %3dx%3d  %.1f us" % (c, h, w, e0.elapsed_time(e1) * 1000 / 30))
PY
run() { echo "$1"; env $1 timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no_cpu_baseline --no_roofline 2>gpurun_out/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('   ', round(d['value'],1), round(d['ms_per_step'],3))"; }
for i in 1 2 3; do run FD_NONE=1; done
