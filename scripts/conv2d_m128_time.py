"""Stand-alone time of the F(2x2, 3x3) slab route (kernel + k_wino2d_finish) with 64 (k_conv_wino2d) and 128 (k_conv_wino2d_m128) output
channels per workgroup: the deep trunk shapes and the decoder's reflect-padded blocks, batch 12 and 24: conv2d_m128_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
shapes = [(128, 128, 24, 80, "zero"), (256, 256, 12, 40, "zero"), (512, 512, 6, 20, "zero"), (512, 256, 6, 20, "reflect"), (512, 256, 12, 40, "reflect"),
          (256, 128, 12, 40, "reflect"), (256, 128, 24, 80, "reflect")]
for B in (12, 24):
    for ci, co, h, w, mode in shapes:
        if mode == "reflect" and B == 24: continue
        ts = []
        for m128 in (1, 0, 2):
            tuning.set_lib(wino_fwd_2d_min=1, wino_fwd_2dp_min_wgs=0, wino_fwd_2d_m128=m128)
            x = torch.randn(B, ci, h, w, device="cuda")
            wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
            wt._fd_cache_id = -7 - ci - 1000 * co - 1000000 * m128 - 31 * h
            run = lambda: FD.conv2d(x, wt, None, 1, 1, mode)
            with torch.no_grad():
                for _ in range(5): run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50): run()
                e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 50)
        flops = 2.0 * B * h * w * ci * co * 9
        print("batch %2d  %3d -> %3d  %3dx%3d %-7s  128-channel tiles %6.1f us (%3.0f TF/s)   64-channel tiles %6.1f us (%3.0f TF/s)   slot-fill rule %6.1f us"
              % (B, ci, co, h, w, mode, ts[0], flops / ts[0] / 1e6, ts[1], flops / ts[1] / 1e6, ts[2]), flush=True)
