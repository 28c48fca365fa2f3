"""Forward-conv tile / split-K sweep on the step's shapes (tuning aid for conv_fast.hip::choose_config).
usage (GPU box): python scripts/conv_cfg_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning

SHAPES = [("layer1", 64, 64, 48, 160), ("layer2", 128, 128, 24, 80), ("layer3", 256, 256, 12, 40), ("layer4", 512, 512, 6, 20),
          ("dec4", 512, 256, 6, 20), ("dec3", 256, 128, 12, 40), ("dec2", 128, 64, 24, 80)]


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


for B in (12, 24):
    for name, ci, co, h, w in SHAPES:
        x = torch.randn(B, ci, h, w, device="cuda")
        wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
        flops = 2.0 * B * h * w * co * ci * 9
        tuning.set_lib(force_cfg=-1, force_splits=1)
        with torch.no_grad():
            t0 = timeit(lambda: FD.conv2d(x, wt, None, 1, 1))
        res = []
        for cfg in (0, 1, 2):
            if (cfg == 0 and co <= 64) or (cfg == 2 and co > 64):
                continue
            for sp in (1, 2, 3, 4, 6, 8, 12):
                tuning.set_lib(force_cfg=cfg, force_splits=sp)
                with torch.no_grad():
                    t = timeit(lambda: FD.conv2d(x, wt, None, 1, 1))
                res.append((t, cfg, sp))
        tuning.set_lib(force_cfg=-1, force_splits=1)
        res.sort()
        print("B=%2d %-7s model %6.1f us %5.1f TF | best %s | top3 %s" % (
            B, name, t0, flops / t0 / 1e6, "cfg%d x%d %6.1f us %5.1f TF" % (res[0][1], res[0][2], res[0][0], flops / res[0][0] / 1e6),
            " ".join("c%dx%d:%.0f" % (c, s_, t) for t, c, s_ in res[:4])))
