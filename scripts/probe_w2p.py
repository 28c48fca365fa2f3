"""k_conv_wino2p alone on the ResNet layer1 shape (64 -> 64 @ 48x160), for rocprofv3 counter passes: probe_w2p.py [batch] [dma 0|1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
tuning.set_lib(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=1, wino_fwd_2dp_dma=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
x = torch.randn(B, 64, 48, 160, device="cuda")
w = torch.randn(64, 64, 3, 3, device="cuda") * 0.05
w._fd_cache_id = -11
with torch.no_grad():
    for _ in range(12):
        y = FD.conv2d(x, w, None, 1, 1)
torch.cuda.synchronize()
print(float(y.abs().mean()))
