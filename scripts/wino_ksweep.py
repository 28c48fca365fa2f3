"""k_conv_wino: time per launch as a function of the GEMM-K length (input channels) at a fixed tile grid - separates the per-tile
fixed cost (prologue: first-chunk latency; epilogue: output transform + stores) from the per-chunk cost of the main loop.
usage (GPU box): python scripts/wino_ksweep.py [batch] [Cout]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from fusiondepth_amd import _lib, tuning

B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
CO = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W = 48, 160


def timeit(fn, n=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tuning.set_lib(wino_target=1)              # no split-K: the tile grid stays 720 x (Cout / 64)
rows = []
for C in (16, 32, 64, 128, 256, 512):
    x = torch.randn(B, C, H, W, device="cuda")
    w = torch.randn(CO, C, 3, 3, device="cuda") * 0.05
    d = _lib.ConvDesc(B, C, H, W, CO, 3, 3, 1, 1, 0, 0, 0)
    y = torch.empty(B, CO, H, W, device="cuda")
    wt = torch.empty(_lib.query("fd_conv3x3_wino_wt_floats", ctypes.byref(d)), device="cuda")
    ws = torch.empty(max(_lib.query("fd_conv3x3_wino_ws_floats", ctypes.byref(d)), 1), device="cuda")
    st = {"r": 0}

    def run():
        _lib.call("fd_conv3x3_wino_fwd", ctypes.byref(d), x.data_ptr(), w.data_ptr(), None, y.data_ptr(), wt.data_ptr(), st["r"], ws.data_ptr(),
                  _lib.stream())
        st["r"] = 1
    t = timeit(run)
    chunks = 3 * C // 16
    tiles = B * H * (W // 2) // 64 * (CO // 64)
    mfma_us = tiles * chunks * 32 * 4 * 64 / (1024 * 2.4e3)        # all SIMDs busy at 2.4 GHz
    rows.append((C, chunks, t))
    print("C %4d: %3d chunks/tile, %5d tiles: %7.1f us  (matrix pipes busy %4.1f %%, %5.2f us per chunk-round)" % (C, chunks, tiles, t, 100 * mfma_us / t, t / chunks))
(c0, k0, t0), (c1, k1, t1) = rows[2], rows[-1]
per_chunk = (t1 - t0) / (k1 - k0)
print("fit between C=%d and C=%d: %.2f us per chunk (ideal %.2f), fixed %.1f us per launch" % (c0, c1, per_chunk, B * H * (W // 2) // 64 * (CO // 64) * 32 * 4 * 64 / (1024 * 2.4e3), t0 - per_chunk * k0))
