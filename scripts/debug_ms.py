"""Debug: case 606 (no automasking) - gT of oracle / per-scale HIP / multi-scale HIP."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import inputs as gin
from oracle import layers as OL, trainer as OT
from fusiondepth_amd import functional as FD
import test_gpu_losspath as T
seed, B, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 606, 2, 64, 96
over = dict(disable_automasking=True) if seed == 606 else {}
opt = OT.default_opt(height=H, width=W, **over)
inp, rng = gin.batch_inputs(seed, B, H, W)
disp0 = gin.disp_pyramid(rng, B, H, W)
poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
def leaves(cuda):
    mk = (lambda t: t.detach().clone().cuda().requires_grad_(True)) if cuda else (lambda t: t.clone().requires_grad_(True))
    return {s: mk(disp0[("disp", s)]) for s in range(4)}, {f: mk(T0[f]) for f in T0}
d_o, T_o = leaves(False)
terms, outs = T._oracle_photo_terms(opt, inp, d_o, T_o, noise)
tot_o = sum(terms[s][0] for s in range(4))
g_o = torch.autograd.grad(tot_o, [T_o[-1], T_o[1]] + [d_o[s] for s in range(4)])
d_a, T_a = leaves(True)
res = T._hip_photo_terms(FD, opt, inp, d_a, T_a, noise, materialize=False)
g_a = torch.autograd.grad(sum(res[s][0] for s in range(4)), [T_a[-1], T_a[1]] + [d_a[s] for s in range(4)])
for rows in (0, 7, 64):
    d_b, T_b = leaves(True)
    photo, si, sel = T._hip_photo_terms_ms(FD, opt, inp, d_b, T_b, noise, rows)
    g_b = torch.autograd.grad(sum(photo), [T_b[-1], T_b[1]] + [d_b[s] for s in range(4)])
    print("rows", rows)
    for k in range(2):
        sc = g_o[k].abs().max()
        print("  gT f%d: max|old-oracle|/sc %.2e  max|ms-oracle|/sc %.2e" % (k, (g_a[k].cpu() - g_o[k]).abs().max() / sc, (g_b[k].cpu() - g_o[k]).abs().max() / sc))
    for s in range(4):
        flips_a = int((res[s][2].cpu().long() != terms[s][2]).sum()); flips_b = int((sel[s].cpu().long() != terms[s][2]).sum())
        ea = (g_a[2 + s].cpu() - g_o[2 + s]).abs(); eb = (g_b[2 + s].cpu() - g_o[2 + s]).abs(); sc = g_o[2 + s].abs().max()
        print("  s%d flips old %d ms %d | d_disp err/sc: old max %.2e sum %.2e ; ms max %.2e sum %.2e ; photo old %.3e ms %.3e" % (
            s, flips_a, flips_b, ea.max() / sc, ea.sum() / g_o[2 + s].abs().sum(), eb.max() / sc, eb.sum() / g_o[2 + s].abs().sum(),
            abs(float(res[s][0]) - float(terms[s][0])) / float(terms[s][0]), abs(float(photo[s]) - float(terms[s][0])) / float(terms[s][0])))
