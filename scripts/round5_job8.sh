#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
python - <<'P'
import numpy as np, torch, sys
sys.path.insert(0, '.')
from fusiondepth_amd import functional as FD
from oracle import evaluate as OE
img = np.random.RandomState(192 + 1242).uniform(0.01, 0.7, (2, 192, 640)).astype(np.float32)
got = FD.resize_linear_cv(torch.from_numpy(img).cuda(), (375, 1242)).cpu().numpy()
w = OE.resize_bilinear(img[0], 375, 1242)
d = np.abs(got[0] - w); print("max diff", d.max(), "count", (d > 0).sum(), "of", d.size, "where", np.argwhere(d > 0)[:5])
P
bash scripts/prof_bench.sh round5s --no_other_configs > /dev/null 2>&1
grep -n "k_bn_relu_pool\|k_maxpool\|k_bn_stats\|k_bn_apply_train\|k_bn_bwd_apply\|k_bn_bwd_reduce\|total kernel" $O/round5s_bench_kernel_stats.md
