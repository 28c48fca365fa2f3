"""One convolution shape: n forward + data-gradient launches (for the PMC passes).  usage: conv_one.py Cin H W Cout K stride pad batch [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
ci, h, w, co, k, st, pd, B = (int(a) for a in sys.argv[1:9]); n = int(sys.argv[9]) if len(sys.argv) > 9 else 8
x = torch.randn(B, ci, h, w, device="cuda").requires_grad_(True)
wt = (torch.randn(co, ci, k, k, device="cuda") * 0.05)
wt._fd_cache_id = -2
for _ in range(n):
    y = FD.conv2d(x, wt, None, st, pd)
    torch.autograd.grad(y, [x], torch.ones_like(y))
torch.cuda.synchronize()
