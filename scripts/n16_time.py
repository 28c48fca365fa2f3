"""Stand-alone time of the decoder's full-resolution 3x3 blocks, forward and data gradient, on conv_n16.hip and on the implicit-GEMM
kernel: n16_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
for cin, cout, h, w in [(16, 16, 192, 640), (32, 16, 96, 320)]:
    for what in ("fwd", "fwd+dgrad"):
        ts = []
        for n16 in ("1", "0"):
            tuning.set_lib(conv_n16_min_pixels=16384 if n16 == "1" else -1)
            x = torch.randn(12, cin, h, w, device="cuda", requires_grad=True)
            wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
            wt._fd_cache_id = -500 - cin - 100 * int(n16)
            b = torch.zeros(cout, device="cuda")
            gy = torch.randn(12, cout, h, w, device="cuda")
            def run():
                y = FD.conv2d(x, wt, b, 1, 1, "reflect", "elu")
                if what != "fwd":
                    torch.autograd.grad(y, [x], gy)
            for _ in range(3): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1000 / 20)
        print("%2d -> %2d  %3dx%3d batch 12  %-9s  n16 %7.1f us   implicit GEMM %7.1f us" % (cin, cout, h, w, what, ts[0], ts[1]))
