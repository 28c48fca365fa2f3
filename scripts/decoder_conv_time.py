"""Stand-alone time of the depth decoder's wide ConvBlocks (reflect padding, bias, ELU; batch 12) on the three Winograd kernels,
forward and forward + data gradient: decoder_conv_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
shapes = [("upconv(4,0)", 512, 256, 6, 20), ("upconv(4,1)", 512, 256, 12, 40), ("upconv(3,0)", 256, 128, 12, 40), ("upconv(3,1)", 256, 128, 24, 80),
          ("upconv(2,0)", 128, 64, 24, 80), ("upconv(2,1)", 128, 64, 48, 160)]
MODES = {"1-D": dict(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=0), "2p": dict(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=1),
         "slabs": dict(wino_fwd_2d_min=1, wino_fwd_2dp_min_wgs=0), "default": dict(wino_fwd_2d_min=65536, wino_fwd_2dp_min_wgs=160)}
B = 12
for name, ci, co, h, w in shapes:
    for what in ("fwd", "fwd+dgrad"):
        ts = {}
        for k, (mode, fields) in enumerate(MODES.items()):
            tuning.set_lib(**fields)
            x = torch.randn(B, ci, h, w, device="cuda", requires_grad=True)
            wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
            wt._fd_cache_id = -9000 - ci - 10 * co - 100000 * k - h
            b = torch.zeros(co, device="cuda")
            gy = torch.randn(B, co, h, w, device="cuda")
            def run():
                y = FD.conv2d(x, wt, b, 1, 1, "reflect", "elu")
                if what != "fwd":
                    torch.autograd.grad(y, [x], gy)
            for _ in range(4): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ts[mode] = e0.elapsed_time(e1) * 1000 / 20
        print("%-12s %3d -> %3d %3dx%3d  %-9s " % (name, ci, co, h, w, what) + "  ".join("%s %6.1f us" % (m, t) for m, t in ts.items()), flush=True)
