"""k_wgrad_wino (+ its finish pass) on the ResNet-18 layer shapes of the training step: time per launch against the MFMA work it
executes (4 products per pixel pair and kernel row instead of 6 = 2/3 of the direct algorithm), for a sweep of the pixel-slice target.
usage (GPU box): python scripts/wgrad_probe.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SHAPES = [("layer1", 64, 64, 48, 160), ("layer2", 128, 128, 24, 80), ("layer3", 256, 256, 12, 40), ("layer4", 512, 512, 6, 20)]


def timeit(fn, n=30):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, ci, co, h, w in SHAPES:
    x = torch.randn(B, ci, h, w, device="cuda")
    gy = torch.randn(B, co, h, w, device="cuda")
    gw = torch.zeros(co, ci, 3, 3, device="cuda")
    d = _lib.ConvDesc(B, ci, h, w, co, 3, 3, 1, 1, 0, 0, 0)
    ws = torch.empty(max(_lib.query("fd_conv2d_bwd_weight_ws_floats", ctypes.byref(d)), 1), device="cuda")

    def run():
        _lib.call("fd_conv2d_bwd_weight", ctypes.byref(d), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), None, ws.data_ptr(), 0, _lib.stream())
    t = timeit(run)
    flops = 2.0 * B * h * w * co * ci * 9
    mfma_us = flops * (2.0 / 3.0) / 157.3e6
    ref = torch.nn.grad.conv2d_weight(x.double(), gw.shape, gy.double(), padding=1) if B * h * w <= 12 * 48 * 160 else None
    err = float((gw.double() - ref).abs().max() / ref.abs().max()) if ref is not None else float("nan")
    print("%-7s %3d->%3d @%dx%d batch %d: %7.1f us per launch incl. finish (matrix-pipe time of its 2/3 products: %5.1f us = %4.1f %%), err %.1e"
          % (name, ci, co, h, w, B, t, mfma_us, 100 * mfma_us / t, err))
