"""Stand-alone time of the depth decoder's 32-output-channel ConvBlocks (upconv(1,0), upconv(1,1): reflect padding, bias, ELU; batch 12)
on the direct kernels (fd_tuning.wino_min_cout / wino_wgrad_min_cout = 64, rounds 1-5) against the Winograd kernels with half of their
64-channel tile idle (= 32): forward, data gradient and weight gradient one by one.  decoder_m32_time.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import functional as FD, tuning
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
shapes = [("upconv(1,0)", 64, 32, 48, 160), ("upconv(1,1)", 96, 32, 96, 320), ("upconv(1,0) 1024x320", 64, 32, 80, 256),
          ("upconv(1,1) 1024x320", 96, 32, 160, 512), ("upconv(2,1)", 128, 64, 48, 160)]
k = 0
for name, ci, co, h, w in shapes:
    for what in ("fwd", "dgrad", "wgrad"):
        ts = {}
        for mc in (64, 32):
            k += 1
            tuning.set_lib(wino_min_cout=mc, wino_wgrad_min_cout=mc)
            x = torch.randn(B, ci, h, w, device="cuda", requires_grad=True)
            wt = (torch.randn(co, ci, 3, 3, device="cuda") * 0.05).requires_grad_(True)
            wt._fd_cache_id = -50000 - k
            b = torch.zeros(co, device="cuda")
            gy = torch.randn(B, co, h, w, device="cuda")
            y = FD.conv2d(x, wt, b, 1, 1, "reflect", "elu")
            def run():
                if what == "fwd":
                    with torch.no_grad():
                        FD.conv2d(x, wt, b, 1, 1, "reflect", "elu")
                else:
                    torch.autograd.grad(y, [x] if what == "dgrad" else [wt], gy, retain_graph=True)
            for _ in range(4): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run()
            e1.record(); torch.cuda.synchronize()
            ts[mc] = e0.elapsed_time(e1) * 1000 / 20
        gf = 2.0 * B * ci * co * 9 * h * w / 1e9
        print("%-22s %3d -> %3d %3dx%3d b%-2d %-6s %5.1f GFLOP  " % (name, ci, co, h, w, B, what, gf) + "  ".join("min_cout %d: %7.1f us" % (m, t) for m, t in ts.items()), flush=True)
