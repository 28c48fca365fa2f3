"""Stand-alone time of the 3x3 forward convolution of ResNet layer1 / layer2 (and the decoder's 128 -> 64 block) on the three Winograd
kernels: F(2, 3) along x (k_conv_wino), F(2x2, 3x3) in one workgroup (k_conv_wino2p, round 4) and F(2x2, 3x3) with row-component
slabs (k_conv_wino2d + finish); batch 12 and 24, cached weight layouts: conv2p_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
shapes = [(64, 64, 48, 160), (128, 128, 24, 80), (128, 64, 48, 160), (256, 256, 12, 40)]
MODES = {"1-D": dict(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=0), "2x2 one workgroup": dict(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=1),
         "2x2 slabs": dict(wino_fwd_2d_min=1, wino_fwd_2dp_min_wgs=0)}
for B in (12, 24):
    for ci, co, h, w in shapes:
        ts = {}
        for k, (name, fields) in enumerate(MODES.items()):
            tuning.set_lib(**fields)
            x = torch.randn(B, ci, h, w, device="cuda")
            wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
            wt._fd_cache_id = -3 - ci - 1000 * co - 100000 * k
            run = lambda: FD.conv2d(x, wt, None, 1, 1)
            with torch.no_grad():
                for _ in range(5): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): run()
                e1.record(); torch.cuda.synchronize()
            ts[name] = e0.elapsed_time(e1) * 1000 / 50
        flops = 2.0 * B * h * w * ci * co * 9
        print("batch %2d  %3d -> %3d  %3dx%3d   " % (B, ci, co, h, w) + "   ".join("%s %6.1f us (%3.0f TF/s)" % (n, t, flops / t / 1e6) for n, t in ts.items()), flush=True)
