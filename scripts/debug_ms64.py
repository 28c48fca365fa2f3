"""Who is closer to a float64 evaluation of the loss path: the fp32 oracle (= the reference's arithmetic), the per-scale HIP
kernels (which mimic that arithmetic) or the multi-scale HIP kernel (centred / scaled SSIM sums)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import inputs as gin
from oracle import layers as OL, trainer as OT
from fusiondepth_amd import functional as FD
import test_gpu_losspath as T
seed, B, H, W = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
opt = OT.default_opt(height=H, width=W)
inp, rng = gin.batch_inputs(seed, B, H, W)
disp0 = gin.disp_pyramid(rng, B, H, W)
poses = {f: gin.small_poses(rng, B) for f in (-1, 1)}
T0 = {f: OL.transformation_from_parameters(*poses[f], invert=(f < 0)) for f in (-1, 1)}
noise = [torch.from_numpy(np.random.RandomState(1000 + seed + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
def run_oracle(dt):
    i2 = {k: (v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in inp.items()}
    d = {s: disp0[("disp", s)].to(dt).clone().requires_grad_(True) for s in range(4)}
    Tq = {f: T0[f].to(dt).clone().requires_grad_(True) for f in T0}
    terms, _ = T._oracle_photo_terms(opt, i2, d, Tq, [n.to(dt) for n in noise])
    tot = sum(terms[s][0] for s in range(4))
    g = torch.autograd.grad(tot, [d[s] for s in range(4)])
    return [float(terms[s][0]) for s in range(4)], [x.double().numpy() for x in g], [terms[s][2].numpy() for s in range(4)]
v64, g64, s64 = run_oracle(torch.float64)
v32, g32, s32 = run_oracle(torch.float32)
def hip(ms):
    d = {s: disp0[("disp", s)].clone().cuda().requires_grad_(True) for s in range(4)}
    Tq = {f: T0[f].clone().cuda().requires_grad_(True) for f in T0}
    if ms:
        photo, si, sel = T._hip_photo_terms_ms(FD, opt, inp, d, Tq, noise)
        sels = [sel[s].cpu().numpy() for s in range(4)]
    else:
        res = T._hip_photo_terms(FD, opt, inp, d, Tq, noise, materialize=False)
        photo = [r[0] for r in res]; sels = [r[2].cpu().numpy() for r in res]
    g = torch.autograd.grad(sum(photo), [d[s] for s in range(4)])
    return [float(p) for p in photo], [x.double().cpu().numpy() for x in g], sels
for name, (v, g, sl) in (("oracle fp32", (v32, g32, s32)), ("HIP per-scale", hip(False)), ("HIP multi-scale", hip(True))):
    print(name)
    for s in range(4):
        sc = np.abs(g64[s]).max()
        err = np.abs(g[s] - g64[s])
        bad = (err > 2e-4 * sc + 2e-3 * np.abs(g64[s])).mean()
        print("  s%d: loss rel err %.2e | argmin flips vs fp64 %5d | grad: %.3f%% out of tol, rel-L1 %.2e, max/sc %.2e" % (
            s, abs(v[s] - v64[s]) / v64[s], int((sl[s].astype(np.int64) != s64[s]).sum()), 100 * bad, err.sum() / np.abs(g64[s]).sum(), err.max() / sc))
