import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD
B = 6
for (cin, cout, h, w) in [(64, 128, 48, 160), (128, 128, 48, 160), (256, 128, 48, 160), (512, 128, 48, 160),
                          (128, 128, 24, 80), (256, 256, 12, 40), (512, 512, 6, 20), (64, 64, 48, 160), (128, 128, 96, 320)]:
    x = torch.randn(B, cin, h, w, device="cuda")
    wt = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    with torch.no_grad():
        for _ in range(3):
            y = FD.conv2d(x, wt, None, 1, 1)
    torch.cuda.synchronize()
print("done")
