"""Stand-alone time of the Winograd forward / data-gradient kernel (+ its split-K reduction) on the four trunk shapes, batch 12
and 24, cached weight layout: conv_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
shapes = [(64, 48, 160), (128, 24, 80), (256, 12, 40), (512, 6, 20)]
for B in (12, 24):
    for c, h, w in shapes:
        ts = []
        for two_d in ("1", "0"):
            tuning.set_lib(wino_fwd_2d_min=1 if two_d == "1" else 0)
            x = torch.randn(B, c, h, w, device="cuda")
            wt = torch.randn(c, c, 3, 3, device="cuda") * 0.05
            wt._fd_cache_id = -2 - c - 10000 * int(two_d)
            run = lambda: FD.conv2d(x, wt, None, 1, 1)
            with torch.no_grad():
                for _ in range(5): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): run()
                e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 50)
        flops = 2.0 * B * h * w * c * c * 9
        print("batch %2d  %3d -> %3d  %3dx%3d   F(2x2,3x3) + finish %6.1f us (%.0f TFLOP/s direct-equivalent)   F(2,3) per row %6.1f us (%.0f)" % (B, c, c, h, w, ts[0], flops / ts[0] / 1e6, ts[1], flops / ts[1] / 1e6))
