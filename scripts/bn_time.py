"""Stand-alone time of the BatchNorm kernels (training forward, backward) on the trunk planes, batch 12 / 24 with grouped statistics,
against the bytes they have to move: bn_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n
for B, groups in ((12, 2), (24, 4)):
    for c, h, w in [(64, 96, 320), (64, 48, 160), (128, 24, 80), (256, 12, 40), (512, 6, 20)]:
        bn = torch.nn.BatchNorm2d(c).cuda().train()
        x = torch.randn(B, c, h, w, device="cuda", requires_grad=True)
        res = torch.randn(B, c, h, w, device="cuda")
        gy = torch.randn(B, c, h, w, device="cuda")
        mb = x.numel() * 4 / 1e6
        with FD.bn_groups(groups):
            tf = t(lambda: FD.batch_norm(x, bn, res, True))
            y = FD.batch_norm(x, bn, res, True)
            tb = t(lambda: torch.autograd.grad(y, [x], gy, retain_graph=True))
        print("batch %2d  %3d ch %3dx%3d (%5.1f MB)  fwd %6.1f us (%4.2f TB/s of 3 reads + 1 write)   bwd %6.1f us (%4.2f TB/s of 5 reads + 1 write... incl. autograd glue)" % (
            B, c, h, w, mb, tf, 4 * mb / tf, tb, 6 * mb / tb))
