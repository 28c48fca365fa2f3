"""Diagnostic (GPU box): are disp-gradient outliers of the fused photometric loss explained by argmin flips?"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import test_gpu_losspath as T
from fusiondepth_amd import functional as FD

for seed, B, H, W in [(404, 2, 64, 96), (505, 1, 192, 640)]:
    opt, terms, outs, res, d_o, T_o, d_g, T_g = T._photo_case(FD, seed, B, H, W)
    for s in (0, 2):
        photo, si, sel, depth, sample, color = res[s]
        want = torch.autograd.grad(terms[s][0], d_o[s], retain_graph=True)[0].numpy()
        got = torch.autograd.grad(photo, d_g[s], retain_graph=True)[0].cpu().numpy()
        selm = (sel.cpu().numpy().astype(np.int64) != terms[s][2].numpy())
        err = np.abs(got - want)
        bad = err > 2e-4 * np.abs(want).max() + 2e-3 * np.abs(want)
        print("seed %d scale %d: sel mismatches %d, outliers %d / %d, agg %.3g" % (seed, s, selm.sum(), bad.sum(), bad.size, err.sum() / np.abs(want).sum()))
        if s == 0:
            ys, xs = np.nonzero(selm.any(0)) if selm.ndim == 3 else np.nonzero(selm)
            mm = np.argwhere(selm)
            for (b, _, y, x) in np.argwhere(bad)[:12]:
                dist = min([max(abs(y - m[1]), abs(x - m[2])) for m in mm if m[0] == b] or [999])
                print("   outlier b%d (%d,%d): got %.4g want %.4g, distance to nearest sel flip %s" % (b, y, x, got[b, 0, y, x], want[b, 0, y, x], dist))
        # SI-only and photo-only split
    gsi_w = torch.autograd.grad(terms[0][1], d_o[0], retain_graph=True)[0].numpy()
    gsi_g = torch.autograd.grad(res[0][1], d_g[0], retain_graph=True)[0].cpu().numpy()
    print("   si grad agg err %.3g" % (np.abs(gsi_g - gsi_w).sum() / np.abs(gsi_w).sum()))
    for f in (-1, 1):
        w = torch.autograd.grad(terms[0][0], T_o[f], retain_graph=True)[0].numpy()
        g = torch.autograd.grad(res[0][0], T_g[f], retain_graph=True)[0].cpu().numpy()
        print("   gT f%d max rel-to-max err %.3g\n%s\n%s" % (f, np.abs(g - w).max() / np.abs(w).max(), g[0], w[0]))
