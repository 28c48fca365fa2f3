"""BatchNorm kernels on the step's shapes: time per launch pair (HIP events around a hipGraph of N launches) and HBM GB/s.
Usage (GPU box): python scripts/bn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn as nn  # noqa: E402

import fusiondepth_amd.functional as FD  # noqa: E402

SHAPES = [(24, 64, 96, 320, 2), (12, 64, 96, 320, 2), (24, 64, 48, 160, 2), (12, 64, 48, 160, 2), (24, 128, 24, 80, 2),
          (24, 256, 12, 40, 2), (24, 512, 6, 20, 2), (12, 512, 6, 20, 2)]


def timed(fn, iters=40):
    """us per call: HIP events around a queue of launches (small shapes are host-bound: read those from rocprofv3)."""
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    print("%-22s %10s %10s %10s %10s" % ("shape (N,C,H,W,G)", "fwd us", "fwd GB/s", "bwd us", "bwd GB/s"))
    tot_f = tot_b = 0.0
    for (N, C, H, W, G) in SHAPES:
        bn = nn.BatchNorm2d(C).cuda()
        x = torch.randn(N, C, H, W, device="cuda")
        res = torch.randn(N, C, H, W, device="cuda")
        FD._BN_GROUPS[0] = G
        xr = x.clone().requires_grad_(True)
        y = FD.batch_norm(xr, bn, residual=res, relu=True)
        gy = torch.randn_like(y)
        nbytes = x.numel() * 4
        t_f = timed(lambda: FD.batch_norm(x, bn, residual=res, relu=True))
        t_b = timed(lambda: torch.autograd.grad(y, xr, gy, retain_graph=True))
        # fwd: stats read x; apply read x + residual, write y = 4 passes.  bwd: reduce reads x, y, gy; apply reads x, y, gy, writes gx, g_res = 8
        print("%-22s %10.1f %10.0f %10.1f %10.0f" % (str((N, C, H, W, G)), t_f, 4 * nbytes / t_f / 1e3, t_b, 8 * nbytes / t_b / 1e3))
        tot_f += t_f; tot_b += t_b
    FD._BN_GROUPS[0] = 1
    print("sum fwd %.1f us, bwd %.1f us" % (tot_f, tot_b))


if __name__ == "__main__":
    main()
