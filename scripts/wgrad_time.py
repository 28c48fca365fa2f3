"""Stand-alone time of Winograd weight-gradient shapes (with FD_LIBFDHIP = an ablation build: where the loop's time goes).  wgrad_time.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib, tuning  # noqa: F401
SHAPES = [(64, 64, 48, 160, 24), (128, 128, 24, 80, 24), (256, 256, 12, 40, 24), (256, 256, 12, 40, 12), (512, 512, 6, 20, 24), (512, 512, 6, 20, 12)]
out = []
for ci, co, h, w, B in SHAPES:
    x = torch.randn(B, ci, h, w, device="cuda"); gy = torch.randn(B, co, h, w, device="cuda"); gw = torch.zeros(co, ci, 3, 3, device="cuda")
    d = _lib.ConvDesc(B, ci, h, w, co, 3, 3, 1, 1, 0, 0, 0)
    ws = torch.empty(max(_lib.query("fd_conv2d_bwd_weight_ws_floats", ctypes.byref(d)), 1), device="cuda")
    def run():
        _lib.call("fd_conv2d_bwd_weight", ctypes.byref(d), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), None, ws.data_ptr(), 0, _lib.stream())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): run()
    e1.record(); torch.cuda.synchronize()
    out.append("%dx%d@%dx%d b%d %6.1f" % (ci, co, h, w, B, e0.elapsed_time(e1) * 1000 / 40))
print("%-10s " % os.path.basename(os.environ.get("FD_LIBFDHIP", "libfdhip.so")).replace("libfdhip", "").replace(".so", "") + " | ".join(out))
