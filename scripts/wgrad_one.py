"""One Winograd weight-gradient shape, n launches (for the PMC passes).  usage: wgrad_one.py Cin Cout H W batch [n]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib, tuning  # noqa: F401  (tuning maps the FD_* variables onto fd_set_tuning at import)
ci, co, h, w, B = (int(a) for a in sys.argv[1:6]); n = int(sys.argv[6]) if len(sys.argv) > 6 else 10
x = torch.randn(B, ci, h, w, device="cuda"); gy = torch.randn(B, co, h, w, device="cuda"); gw = torch.zeros(co, ci, 3, 3, device="cuda")
d = _lib.ConvDesc(B, ci, h, w, co, 3, 3, 1, 1, 0, 0, 0)
ws = torch.empty(max(_lib.query("fd_conv2d_bwd_weight_ws_floats", ctypes.byref(d)), 1), device="cuda")
for _ in range(n):
    _lib.call("fd_conv2d_bwd_weight", ctypes.byref(d), x.data_ptr(), gy.data_ptr(), gw.data_ptr(), None, ws.data_ptr(), 0, _lib.stream())
torch.cuda.synchronize()
