"""Stand-alone time of the four 7x7 stride-2 stems' forward (k_conv7s2_stem): 6 / 4 channels at batch 24, 3 / 2 channels at batch 12.  stem_time.py [H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning  # noqa: F401
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (192, 640)
out = []
for B, C in ((24, 6), (24, 4), (12, 3), (12, 2)):
    x = torch.rand(B, C, H, W, device="cuda")
    w = torch.randn(64, C, 7, 7, device="cuda") * 0.05
    def run():
        with torch.no_grad():
            FD.conv2d(x, w, None, 2, 3, "zero", "none")
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(40): run()
    e1.record(); torch.cuda.synchronize()
    gf = 2.0 * B * 64 * C * 49 * (H // 2) * (W // 2) / 1e9
    out.append("b%d %dch %6.1f us (%.0f TFLOP/s)" % (B, C, e0.elapsed_time(e1) * 1000 / 40, gf / (e0.elapsed_time(e1) / 40) / 1e0))
print("%-8s %dx%d  " % (os.path.basename(os.environ.get("FD_LIBFDHIP", "libfdhip.so")).replace("libfdhip", "").replace(".so", "") or "new", H, W) + " | ".join(out))
