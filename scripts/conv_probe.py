"""Micro-benchmark of single conv shapes (GPU box).  usage: conv_probe.py [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 6
SHAPES = [  # name, Cin, H, W, Cout, K, stride, pad, mode
    ("layer1 3x3 64->64 @48x160", 64, 48, 160, 64, 3, 1, 1, "zero"),
    ("layer2 3x3 128->128 @24x80", 128, 24, 80, 128, 3, 1, 1, "zero"),
    ("layer3 3x3 256->256 @12x40", 256, 12, 40, 256, 3, 1, 1, "zero"),
    ("layer4 3x3 512->512 @6x20", 512, 6, 20, 512, 3, 1, 1, "zero"),
    ("stem 7x7 3->64 s2 @192x640", 3, 192, 640, 64, 7, 2, 3, "zero"),
    ("upconv(0,1) 16->16 @192x640 refl", 16, 192, 640, 16, 3, 1, 1, "reflect"),
    ("upconv(1,1) 96->32 @96x320 refl", 96, 96, 320, 32, 3, 1, 1, "reflect"),
    ("layer2.0 3x3 s2 64->128 @48x160", 64, 48, 160, 128, 3, 2, 1, "zero"),
    ("layer3.0 3x3 s2 128->256 @24x80", 128, 24, 80, 256, 3, 2, 1, "zero"),
    ("layer4.0 3x3 s2 256->512 @12x40", 256, 12, 40, 512, 3, 2, 1, "zero"),
    ("layer2.0 down 1x1 s2 64->128 @48x160", 64, 48, 160, 128, 1, 2, 0, "zero"),
    ("upconv(1,0) 64->32 @48x160 refl", 64, 48, 160, 32, 3, 1, 1, "reflect"),
    ("upconv(2,1) 128->64 @48x160 refl", 128, 48, 160, 64, 3, 1, 1, "reflect"),
    ("upconv(4,0) 512->256 @6x20 refl", 512, 6, 20, 256, 3, 1, 1, "reflect"),
]
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for name, ci, h, w, co, k, st, pd, mode in SHAPES:
    x = torch.randn(B, ci, h, w, device="cuda").requires_grad_(True)
    wt = (torch.randn(co, ci, k, k, device="cuda") * 0.05).requires_grad_(True)
    innorm = False      # the encoder normalises its input in a separate pass (fd_input_normalize)
    with torch.no_grad():
        tf = timeit(lambda: FD.conv2d(x, wt, None, st, pd, mode, "none", innorm), iters)
    y = FD.conv2d(x, wt, None, st, pd, mode, "none", innorm)
    gy = torch.randn_like(y)
    ho, wo = y.shape[2:]
    flops = 2.0 * B * ho * wo * co * ci * k * k
    tb = timeit(lambda: torch.autograd.grad(y, [x, wt], gy, retain_graph=True), iters)
    y2 = FD.conv2d(x.detach(), wt, None, st, pd, mode, "none", innorm)      # graph without a data-gradient branch
    tw = timeit(lambda: torch.autograd.grad(y2, [wt], gy, retain_graph=True), iters)
    print("%-36s fwd %7.1f us %5.1f TF | dgrad+wgrad %7.1f us %5.1f TF | wgrad %7.1f us %5.1f TF" % (
        name, tf * 1e3, flops / tf / 1e9, tb * 1e3, 2 * flops / tb / 1e9, tw * 1e3, flops / tw / 1e9))
