"""Stand-alone time of the data gradients of the ResNet stages' stride-2 3x3 convolutions (layerN.0.conv1) and 1x1 downsample convolutions,
the four output-parity classes in one grouped launch: 64x128 tiles (fd_tuning.grp_tile64_below = 0) against 64x64 tiles.  s2_dgrad_time.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
k = 0
for B in (12, 24):
    for name, ci, co, h, w, ks in [("layer2.0.conv1", 64, 128, 48, 160, 3), ("layer3.0.conv1", 128, 256, 24, 80, 3), ("layer4.0.conv1", 256, 512, 12, 40, 3),
                                   ("layer2.0.down", 64, 128, 48, 160, 1), ("layer3.0.down", 128, 256, 24, 80, 1), ("layer4.0.down", 256, 512, 12, 40, 1)]:
        ts = {}
        for thr in (0, 100000):
            k += 1
            tuning.set_lib(grp_tile64_below=thr)
            x = torch.randn(B, ci, h, w, device="cuda", requires_grad=True)
            wt = (torch.randn(co, ci, ks, ks, device="cuda") * 0.05)
            wt._fd_cache_id = -70000 - k
            y = FD.conv2d(x, wt, None, 2, ks // 2, "zero", "none")
            gy = torch.randn_like(y)
            def run():
                torch.autograd.grad(y, [x], gy, retain_graph=True)
            for _ in range(4): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): run()
            e1.record(); torch.cuda.synchronize()
            ts[thr] = e0.elapsed_time(e1) * 1000 / 30
        gf = 2.0 * B * ci * co * ks * ks * (h // 2) * (w // 2) / 1e9
        print("%-16s b%-2d %3d -> %3d %3dx%3d  %5.2f GFLOP (%5.1f us at the fp32 MFMA peak)  64x128: %6.1f us   64x64: %6.1f us" %
              (name, B, ci, co, h, w, gf, gf / 157.3e3 * 1e6, ts[0], ts[100000]), flush=True)
