"""Stand-alone time of the four encoder stems (7x7 stride 2, forward) on conv_stem.hip and on the gather GEMM it replaces: stem_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
for B, C in ((12, 3), (12, 2), (24, 6), (24, 4)):
    ts = []
    for on in (1, 0):
        tuning.set_lib(stem7=on)
        x = torch.randn(B, C, 192, 640, device="cuda")
        w = torch.randn(64, C, 7, 7, device="cuda") * 0.05
        run = lambda: FD.conv2d(x, w, None, 2, 3)
        with torch.no_grad():
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000 / 30)
    flops = 2.0 * B * 96 * 320 * 64 * 49 * C
    print("batch %2d  %d -> 64  192x640   patch kernel %6.1f us (%3.0f TF/s)   gather GEMM %6.1f us (%3.0f TF/s)" % (B, C, ts[0], flops / ts[0] / 1e6, ts[1], flops / ts[1] / 1e6), flush=True)
# weight gradient (k_wgrad_stem + its slab reduction)
tuning.set_lib(stem7=1)
for B, C in ((12, 3), (12, 2), (24, 6), (24, 4)):
    x = torch.randn(B, C, 192, 640, device="cuda")
    w = (torch.randn(64, C, 7, 7, device="cuda") * 0.05).requires_grad_(True)
    gy = torch.randn(B, 64, 96, 320, device="cuda")
    y = FD.conv2d(x, w, None, 2, 3)
    run = lambda: torch.autograd.grad(y, w, gy, retain_graph=True)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1000 / 30
    flops = 2.0 * B * 96 * 320 * 64 * 49 * C
    print("batch %2d  %d -> 64  192x640   weight gradient %6.1f us (%3.0f TF/s)" % (B, C, t, flops / t / 1e6), flush=True)
