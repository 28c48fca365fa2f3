"""Oracle wrappers for the 2-channel LiDAR scatter (reference gen2channel.py:60-117).

``scatter_2channel_c``   — the sequential C restatement (``scatter.c``), bit-exact w.r.t. the reference.
``scatter_2channel_np``  — an order-free numpy formulation ("mean of the highest-confidence donors",
                           donors summed in raster order) used to cross-check the C code.
TEST INFRASTRUCTURE ONLY.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle_scatter.so")

# ROI of LiDAR returns that donate: rows 76..189, cols 2..637 (gen2channel.py:64-65)
ROI_192x640 = (76, 190, 2, 638)


def build(force=False):
    src = os.path.join(_HERE, "scatter.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, src])
    return _SO


def _lib():
    lib = ctypes.CDLL(build())
    f = lib.fd_oracle_scatter_2channel
    f.restype = ctypes.c_int
    f.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int] * 7
    return f


def scatter_2channel_c(beam, roi=ROI_192x640, expand=2):
    """beam: [H,W] float32 array -> (expanded_depth[H,W], confidence[H,W])."""
    beam = np.ascontiguousarray(beam, dtype=np.float32)
    H, W = beam.shape
    depth = np.empty_like(beam)
    conf = np.empty_like(beam)
    cnt = np.empty_like(beam)
    rc = _lib()(beam.ctypes.data, depth.ctypes.data, conf.ctypes.data, cnt.ctypes.data, H, W, *roi, expand)
    if rc != 0:
        raise ValueError("ROI + expand leaves the image")
    return depth, conf


def donor_offsets(expand=2):
    """[(drow, dcol, confidence)] from the TARGET cell to each possible donor, raster order of donor."""
    offs = []
    for dis in range(1, expand + 1):
        for a in range(1, dis + 1):
            b = dis - a
            for dr in (-a, a):
                for dc in ((-b, b) if b else (0,)):
                    offs.append((dr, dc, np.float32(1.0 / (dis + 1))))
    offs.sort(key=lambda t: (t[0], t[1]))
    return offs


def scatter_2channel_np(beam, roi=ROI_192x640, expand=2):
    beam = np.asarray(beam, dtype=np.float32)
    H, W = beam.shape
    r0, r1, c0, c1 = roi
    src = np.zeros_like(beam)
    src[r0:r1, c0:c1] = beam[r0:r1, c0:c1]
    depth = np.zeros_like(beam)
    conf = np.zeros_like(beam)
    offs = donor_offsets(expand)
    levels = sorted({float(c) for _, _, c in offs}, reverse=True)
    own = src != 0
    depth[own] = src[own]
    conf[own] = 1.0
    done = own.copy()
    for lv in levels:
        acc = np.zeros_like(beam)
        cnt = np.zeros_like(beam)
        for dr, dc, c in offs:
            if float(c) != lv:
                continue
            shifted = np.zeros_like(beam)          # shifted[r,c] = src[r+dr, c+dc]
            rs0, rs1 = max(0, -dr), min(H, H - dr)
            cs0, cs1 = max(0, -dc), min(W, W - dc)
            shifted[rs0:rs1, cs0:cs1] = src[rs0 + dr:rs1 + dr, cs0 + dc:cs1 + dc]
            m = shifted != 0
            acc = np.where(m, acc + shifted, acc).astype(np.float32)
            cnt += m
        take = (cnt > 0) & ~done
        depth[take] = (acc[take] / cnt[take]).astype(np.float32)
        conf[take] = np.float32(lv)
        done |= take
    return depth, conf
