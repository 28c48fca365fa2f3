"""1-D Winograd F(2,3) conv kernel vs the direct implicit-GEMM kernel: error against a float64 reference and time per launch.
usage (GPU box): python scripts/wino_probe.py [batch]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from fusiondepth_amd import _lib
from fusiondepth_amd import functional as FD

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
SHAPES = [("layer1", 64, 64, 48, 160, "zero"), ("layer2", 128, 128, 24, 80, "zero"), ("layer3", 256, 256, 12, 40, "zero"),
          ("layer4", 512, 512, 6, 20, "zero"), ("dec 96->32 refl", 96, 32, 96, 320, "reflect"), ("dec 16->16 refl", 16, 16, 192, 640, "reflect"),
          ("dec 512->256 refl", 512, 256, 6, 20, "reflect"), ("odd rows 64->80", 64, 80, 7, 10, "zero")]


def timeit(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def wino(x, w, bias, mode, act=0):
    N, C, H, W = x.shape
    d = _lib.ConvDesc(N, C, H, W, w.shape[0], 3, 3, 1, 1, 1 if mode == "reflect" else 0, act, 0)
    y = torch.empty(N, w.shape[0], H, W, device="cuda")
    wt = torch.empty(_lib.query("fd_conv3x3_wino_wt_floats", ctypes.byref(d)), device="cuda")
    ws_n = _lib.query("fd_conv3x3_wino_ws_floats", ctypes.byref(d))
    ws = torch.empty(max(ws_n, 1), device="cuda")
    state = {"ready": 0}

    def run():
        _lib.call("fd_conv3x3_wino_fwd", ctypes.byref(d), x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                  y.data_ptr(), wt.data_ptr(), state["ready"], ws.data_ptr(), _lib.stream())
        state["ready"] = 1
        return y
    return run


for name, ci, co, h, w_, mode in SHAPES:
    torch.manual_seed(0)
    x = torch.randn(B, ci, h, w_, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    bias = torch.randn(co, device="cuda")
    xp = F.pad(x.double(), (1, 1, 1, 1), mode="reflect" if mode == "reflect" else "constant")
    ref = F.conv2d(xp, wt.double(), bias.double())
    run = wino(x, wt, bias, mode)
    yw = run().double()
    with torch.no_grad():
        yd = FD.conv2d(x, wt, bias, 1, 1, mode).double()
    sc = ref.abs().max()
    ew, ed = float((yw - ref).abs().max() / sc), float((yd - ref).abs().max() / sc)
    flops = 2.0 * B * h * w_ * co * ci * 9
    tw = timeit(run)
    with torch.no_grad():
        td = timeit(lambda: FD.conv2d(x, wt, bias, 1, 1, mode))
    print("%-18s err wino %.2e direct %.2e | wino %6.1f us (%5.1f eff. TF) direct %6.1f us (%5.1f TF) | x%.2f" % (
        name, ew, ed, tw, flops / tw / 1e6, td, flops / td / 1e6, td / tw))
