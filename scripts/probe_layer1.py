"""The bench.py roofline probe kernel on its own (for rocprofv3 --pmc passes): layer1 3x3 conv 64->64 @48x160, batch 12."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = torch.randn(B, 64, 48, 160, device="cuda")
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.03
w._fd_cache_id = -1
with torch.no_grad():
    for _ in range(20):
        y = FD.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
print("done", float(y.abs().mean()))
