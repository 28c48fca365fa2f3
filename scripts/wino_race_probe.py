"""Stress check of the direct-to-LDS (LDS-DMA) Winograd convolution: repeated launches must be bit-identical, and identical to the
register-staged variant (same arithmetic, FD_WINO_DMA=0 in a child process), with a second stream hammering HBM meanwhile.
usage (GPU box): python scripts/wino_race_probe.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib

SHAPES = [(12, 64, 64, 48, 160, 0), (24, 64, 64, 48, 160, 0), (12, 128, 128, 24, 80, 0), (12, 256, 256, 12, 40, 0), (12, 512, 512, 6, 20, 0),
          (12, 512, 256, 12, 40, 1), (12, 128, 64, 48, 160, 1), (3, 64, 80, 8, 12, 0), (2, 32, 64, 20, 36, 1)]


def conv(x, w, b, mode, act=0):
    N, C, H, W = x.shape
    d = _lib.ConvDesc(N, C, H, W, w.shape[0], 3, 3, 1, 1, mode, act, 0)
    y = torch.empty(N, w.shape[0], H, W, device="cuda")
    wt = torch.empty(_lib.query("fd_conv3x3_wino_wt_floats", ctypes.byref(d)), device="cuda")
    ws = torch.empty(max(_lib.query("fd_conv3x3_wino_ws_floats", ctypes.byref(d)), 1), device="cuda")
    _lib.call("fd_conv3x3_wino_fwd", ctypes.byref(d), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), wt.data_ptr(), 0, ws.data_ptr(), _lib.stream())
    return y


def outputs():
    outs = []
    for i, (N, ci, co, h, w_, mode) in enumerate(SHAPES):
        g = torch.Generator(device="cuda").manual_seed(100 + i)
        x = torch.randn(N, ci, h, w_, device="cuda", generator=g)
        w = torch.randn(co, ci, 3, 3, device="cuda", generator=g) * 0.05
        b = torch.randn(co, device="cuda", generator=g)
        outs.append((x, w, b, mode))
    return outs


if len(sys.argv) > 1 and sys.argv[1] == "child":
    res = [conv(*a).cpu() for a in outputs()]
    torch.save(res, sys.argv[2])
    sys.exit(0)

args = outputs()
first = [conv(*a) for a in args]
torch.cuda.synchronize()
bg = torch.cuda.Stream()
junk = torch.randn(64 << 20, device="cuda")
bad = 0
for rep in range(60):
    with torch.cuda.stream(bg):
        for _ in range(3):
            junk.mul_(1.0001)
    for i, a in enumerate(args):
        y = conv(*a)
        if not torch.equal(y, first[i]):
            bad += 1
            print("rep %d shape %s: %d elements differ" % (rep, SHAPES[i], int((y != first[i]).sum())))
torch.cuda.synchronize()
print("repeat launches: %d mismatching of %d" % (bad, 60 * len(args)))
path = "/tmp/wino_nodma.pt"
env = dict(os.environ, FD_WINO_DMA="0")
subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", path], env=env)
ref = torch.load(path)
for i, r in enumerate(ref):
    same = torch.equal(first[i].cpu(), r)
    print("shape %s: LDS-DMA == register-staged: %s%s" % (SHAPES[i], same, "" if same else " (max |diff| %.3g)" % float((first[i].cpu() - r).abs().max())))
