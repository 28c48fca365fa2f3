"""k_conv_wino2d_limb against k_conv_wino2d_m128: forward and data gradient of the layers the F(2x2, 3x3) slab kernels take in the ResNet-18
step at 640x192 (layer3 / layer4, stacked batch 12 / 24), stand-alone time incl. k_wino2d_finish (hipGraph replay), through the C ABI with
cached weight layouts.    python scripts/wino2d_limb_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
from fusiondepth_amd._lib import call, ptr, stream
from limb_ab import timed

SHAPES = [(256, 256, 12, 40, 12), (256, 256, 12, 40, 24), (512, 512, 6, 20, 12), (512, 512, 6, 20, 24)]
for limb in (0, 1):
    tuning.set_lib(wino_fwd_limb=limb)
    out = []
    for ci, co, h, w, B in SHAPES:
        x = torch.randn(B, ci, h, w, device="cuda"); wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.03
        gy = torch.randn(B, co, h, w, device="cuda")
        plan = FD._conv_plan(x, wt, 1, 1, 0, 0, False)
        dp = plan.dp
        y = torch.empty(B, co, h, w, device="cuda"); gx = torch.empty_like(x)
        f_ws = torch.empty(max(plan.fwd_ws, 1), device="cuda"); f_wt = torch.empty(max(plan.fwd_wt, 1), device="cuda")
        d_ws_n, d_wt_n = plan.data_sizes()
        d_ws = torch.empty(max(d_ws_n, 1), device="cuda"); d_wt = torch.empty(max(d_wt_n, 1), device="cuda")
        call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 0, ptr(f_ws), stream())
        call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 0, ptr(d_ws), stream())
        torch.cuda.synchronize()
        t_f = timed(lambda: call("fd_conv2d_fwd", dp, ptr(x), ptr(wt), None, ptr(y), ptr(f_wt), 1, ptr(f_ws), stream()))
        t_d = timed(lambda: call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(wt), ptr(gx), ptr(d_wt), 1, ptr(d_ws), stream()))
        out.append("%dx%d@%dx%d b%d fwd %5.1f dgrad %5.1f" % (ci, co, h, w, B, t_f, t_d))
    print(("limb: " if limb else "f32:  ") + " | ".join(out), flush=True)
tuning.set_lib(wino_fwd_limb=0)
