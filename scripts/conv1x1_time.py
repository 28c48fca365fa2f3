"""Stand-alone times of the ResNet-50 bottleneck's 1x1 convolutions (640x192: layer1 .. layer4 planes), forward / data gradient /
weight gradient, batch 8 and 16, cached weight layouts: which of them sit how far from the MFMA and the HBM floor.
    python scripts/conv1x1_time.py [batch ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD

SHAPES = [  # Cin, Cout, H, W (input plane), stride
    (64, 64, 48, 160, 1), (64, 256, 48, 160, 1), (256, 64, 48, 160, 1), (256, 128, 48, 160, 1),
    (128, 512, 24, 80, 1), (512, 128, 24, 80, 1), (256, 512, 48, 160, 2), (512, 256, 24, 80, 1),
    (256, 1024, 12, 40, 1), (1024, 256, 12, 40, 1), (512, 1024, 24, 80, 2), (1024, 512, 12, 40, 1),
    (512, 2048, 6, 20, 1), (2048, 512, 6, 20, 1), (1024, 2048, 12, 40, 2)]


def timed(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / n


for B in ([int(a) for a in sys.argv[1:]] or [8, 16]):
    tot = [0.0, 0.0, 0.0]
    for k, (ci, co, h, w, st) in enumerate(SHAPES):
        x = torch.randn(B, ci, h, w, device="cuda").requires_grad_(True)
        wt = (torch.randn(co, ci, 1, 1, device="cuda") * 0.05).requires_grad_(True)
        wt._fd_cache_id = -100 - k - 1000 * B
        y = FD.conv2d(x, wt, None, st, 0)
        gy = torch.randn_like(y)
        with torch.no_grad():
            t_f = timed(lambda: FD.conv2d(x, wt, None, st, 0))
        xd = x.detach().requires_grad_(True)
        wd = wt.detach(); wd._fd_cache_id = wt._fd_cache_id
        yd = FD.conv2d(xd, wd, None, st, 0)
        t_d = timed(lambda: torch.autograd.grad(yd, [xd], gy, retain_graph=True))
        xw = x.detach()
        yw = FD.conv2d(xw, wt, None, st, 0)
        t_w = timed(lambda: torch.autograd.grad(yw, [wt], gy, retain_graph=True))
        ho, wo = y.shape[2:]
        flops = 2.0 * B * ho * wo * ci * co
        byts = 4.0 * (x.numel() + y.numel() + wt.numel())
        floor = max(flops / 157.3e12, byts / 8e12) * 1e6
        tot[0] += t_f; tot[1] += t_d; tot[2] += t_w
        print("b%-2d %4d -> %4d @%3dx%3d s%d  %6.2f GFLOP %6.1f MB floor %5.1f us | fwd %6.1f us (%5.1f TF, %4.2f TB/s)  dgrad %6.1f (%5.1f TF)  wgrad %6.1f (%5.1f TF)"
              % (B, ci, co, h, w, st, flops / 1e9, byts / 1e6, floor, t_f, flops / t_f / 1e6, byts / t_f / 1e6, t_d, flops / t_d / 1e6, t_w, flops / t_w / 1e6), flush=True)
    print("b%-2d sum: fwd %.0f us, dgrad %.0f us, wgrad %.0f us" % (B, *tot))
