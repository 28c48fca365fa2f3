"""Stand-alone time of the Winograd weight gradient (kernel + slab reduction) on the four trunk shapes, batch 12 and 24, with the
2-D and the 1-D algorithm: wgrad_time.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib, tuning
shapes = [(64, 64, 48, 160), (128, 128, 24, 80), (256, 256, 12, 40), (512, 512, 6, 20)]
for B in (12, 24):
    for ci, co, h, w in shapes:
        row = []
        for two_d in ("1", "0"):
            tuning.set_lib(wino_wgrad_2d=int(two_d))
            x = torch.randn(B, ci, h, w, device="cuda"); gy = torch.randn(B, co, h, w, device="cuda"); gw = torch.zeros(co, ci, 3, 3, device="cuda")
            d = _lib.ConvDesc(B, ci, h, w, co, 3, 3, 1, 1, 0, 0, 0)
            ws = torch.empty(max(_lib.query("fd_conv2d_bwd_weight_ws_floats", ctypes.byref(d)), 1), device="cuda")
            run = lambda: _lib.call("fd_conv2d_bwd_weight", ctypes.byref(d), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), None, ws.data_ptr(), 0, _lib.stream())
            for _ in range(5): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) * 1000 / 50)
        flops = 2.0 * B * h * w * ci * co * 9
        print("batch %2d  %3d -> %3d  %3dx%3d   2-D %6.1f us (%.0f TFLOP/s direct-equivalent)   1-D %6.1f us" % (B, ci, co, h, w, row[0], flops / row[0] / 1e6, row[1]))
