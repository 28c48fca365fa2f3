"""Beam-encoder stem on a sparse LiDAR map: which of (input normalise, 7x7 conv, BatchNorm) loses accuracy?  Development tool."""
import os, sys
import numpy as np, torch
import torch.nn.functional as F
here = os.path.dirname(os.path.abspath(__file__))
for p in ("..", "../tests", "../tests/golden"):
    sys.path.insert(0, os.path.join(here, p))
import test_gpu_trainer as T
from fusiondepth_amd import functional as FD

H, W, B = 352, 1216, 1
opt = T._opts(height=H, width=W, batch_size=B)
tr, ot = T._make_pair(opt)
inp, _ = T._batch(B, H, W, 900)
two = inp["2channel"]
eo = ot.models["beam_encoder"].encoder
eg = tr.models["beam_encoder"].encoder
for m in (eo, eg):
    m.train()
w = eo.conv1.weight.detach()


def e(a, r):
    a, r = a.detach().double().cpu(), r.detach().double().cpu()
    return float((a - r).abs().max() / r.pow(2).mean().sqrt())


with torch.no_grad():
    xn64 = (two.double() - 0.45) / 0.225
    y64 = F.conv2d(xn64, w.double(), None, 2, 3)
    y32 = F.conv2d(((two - 0.45) / 0.225), w, None, 2, 3)
    xg = FD.input_normalize(two.cuda())
    print("input_normalize err", e(xg, xn64))
    yg = FD.conv2d(xg, eg.conv1.weight, None, stride=2, pad=3)
    sd = y64.std((0, 2, 3)); mu = y64.mean((0, 2, 3))
    print("conv1 out: |mean|/std per channel: median %.1f max %.1f ; std median %.3g" % (float((mu.abs() / sd).median()), float((mu.abs() / sd).max()), float(sd.median())))
    # conv error in units of the channel's std (what BatchNorm turns it into)
    for name, y in (("torch f32", y32), ("HIP", yg.cpu())):
        d = (y.double() - y64).abs() / sd.view(1, -1, 1, 1)
        print("conv1 %-9s max |d|/std_c %.3e   max|d|/rms %.3e" % (name, float(d.max()), e(y, y64)))
    import copy
    bn64 = copy.deepcopy(eo.bn1).double()
    o64 = F.relu(bn64(y64))
    bn32 = copy.deepcopy(eo.bn1)
    print("BN torch f32 on f32(y64): %.3e" % e(F.relu(bn32(y64.float())), o64))
    bng = copy.deepcopy(eo.bn1).cuda()
    print("BN HIP       on f32(y64): %.3e" % e(FD.batch_norm(y64.float().cuda(), bng, relu=True), o64))
    bng = copy.deepcopy(eo.bn1).cuda()
    print("BN HIP on HIP conv      : %.3e" % e(FD.batch_norm(yg, bng, relu=True), o64))
    bn32 = copy.deepcopy(eo.bn1)
    print("BN torch f32 on HIP conv: %.3e" % e(F.relu(bn32(yg.cpu())), o64))
    bn32 = copy.deepcopy(eo.bn1)
    print("BN torch f32 on torch f32 conv: %.3e" % e(F.relu(bn32(y32)), o64))
    # statistics
    m64 = y64.mean((0, 2, 3)); v64 = y64.var((0, 2, 3), unbiased=False)
    bng = copy.deepcopy(eo.bn1).cuda(); bng.momentum = 1.0
    FD.batch_norm(y64.float().cuda(), bng, relu=True)
    n = y64.numel() / y64.shape[1]
    print("HIP running_mean rel err (vs std): %.3e   running_var rel err %.3e" % (
        float(((bng.running_mean.cpu().double() - m64).abs() / sd).max()),
        float(((bng.running_var.cpu().double() - v64 * n / (n - 1)).abs() / v64).max())))
