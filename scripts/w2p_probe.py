"""k_conv_wino2p alone: time per launch on the layer1 / layer2 shapes (batch 12 / 24), random data.  Run with FD_LIBFDHIP=<ablation build>
to see what a loop ingredient costs (scripts/build_ablation.sh w2p_<tag> conv_wino -DFD_W2P_ABLATE=<bits>): w2p_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
tuning.set_lib(wino_fwd_2d_min=0, wino_fwd_2dp_min_wgs=1)
out = []
for B, ci, co, h, w in ((12, 64, 64, 48, 160), (24, 64, 64, 48, 160), (12, 128, 128, 24, 80), (24, 128, 128, 24, 80), (36, 64, 64, 48, 160)):
    x = torch.randn(B, ci, h, w, device="cuda")
    wt = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wt._fd_cache_id = -7 - ci - 1000 * B
    run = lambda: FD.conv2d(x, wt, None, 1, 1)
    with torch.no_grad():
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1000 / 50
    wgs = B * (h // 2) * (w // 2) // 64 * (co // 64)
    mfma_us = wgs * 4 * (ci // 16) * 32 * 4 * 64 / (1024 * 2.4e3)
    out.append("b%d %d@%dx%d %5.1f us (%d WGs, pipes %2.0f %%)" % (B, ci, h, w, t, wgs, 100 * mfma_us / t))
print("%-12s " % os.environ.get("FD_LIBFDHIP", "base").split("libfdhip")[-1] + " | ".join(out), flush=True)
