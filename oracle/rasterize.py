"""Oracle restatement of the LiDAR rasterisation upstream of the 2-channel scatter.  TEST INFRASTRUCTURE ONLY.

Reference: kitti_utils.py:40-102 ``generate_depth_map`` (Velodyne points -> z-buffered sparse depth image in camera 2) and
datasets/kitti_dataset.py:93-117 ``get_4beam`` + mono_dataset.py:193-198 (pad to 384x1280, 2x2 max-pool with ceil_mode,
float32, / 100 -> the "4beam" input).  Restated in numpy float64 exactly as the reference computes (projection, np.round =
round-half-even, the "- 1" MATLAB offset, bounds test) with the duplicate handling written out as what it *does*:

* a pixel first receives the depth of the LAST point that lands on it (fancy-index assignment, kitti_utils.py:76);
* duplicates are found through ``sub2ind`` = row * (W - 1) + col - 1 (kitti_utils.py:33-37: W - 1, not W), so besides points
  on one pixel, the first column of row r and the last column of row r - 1 share an index; every index with more than one
  point writes the MINIMUM depth of its points to the pixel of its FIRST point (kitti_utils.py:80-85).
"""
import numpy as np


def project_points(velo, P_velo2im, im_h, im_w, vel_depth=False):
    """kitti_utils.py:59-72: -> (col, row, depth) of the points that land inside the image, in point order."""
    velo = velo[velo[:, 0] >= 0, :].copy()
    velo[:, 3] = 1.0
    pts = np.dot(P_velo2im, velo.T).T
    pts[:, :2] = pts[:, :2] / pts[:, 2][..., np.newaxis]
    if vel_depth:                                         # kitti_utils.py:68-69
        pts[:, 2] = velo[:, 0]
    pts[:, 0] = np.round(pts[:, 0]) - 1
    pts[:, 1] = np.round(pts[:, 1]) - 1
    ok = (pts[:, 0] >= 0) & (pts[:, 1] >= 0) & (pts[:, 0] < im_w) & (pts[:, 1] < im_h)
    pts = pts[ok]
    return pts[:, 0].astype(np.int64), pts[:, 1].astype(np.int64), pts[:, 2]


def depth_image(velo, P_velo2im, im_h, im_w, vel_depth=False):
    """kitti_utils.py:74-86: sparse depth [im_h, im_w] (float64)."""
    col, row, z = project_points(velo, P_velo2im, im_h, im_w, vel_depth)
    depth = np.zeros((im_h, im_w))
    depth[row, col] = z                                   # last point wins
    inds = row * (im_w - 1) + col - 1                     # the reference's sub2ind
    order = np.argsort(inds, kind="stable")
    si = inds[order]
    start = np.flatnonzero(np.r_[True, si[1:] != si[:-1]])
    stop = np.r_[start[1:], si.size]
    for a, b in zip(start, stop):
        if b - a > 1:
            grp = order[a:b]                              # ascending point index (stable sort)
            depth[row[grp[0]], col[grp[0]]] = z[grp].min()
    depth[depth < 0] = 0
    return depth


def pad_to_shape(depth, shape):
    """kitti_utils.py:88-101 (``shape`` branch)."""
    crop = shape[0] < depth.shape[0]
    ypad = abs(shape[0] - depth.shape[0])
    xpad = shape[1] - depth.shape[1]
    xpad1 = xpad // 2
    depth = np.pad(depth, ((ypad, 0), (xpad1, xpad - xpad1)))
    if crop:
        depth = depth[2:, :]
    return depth


def max_pool2x2_ceil(a):
    """F.max_pool2d(x, 2, ceil_mode=True) on a 2-D array."""
    H, W = a.shape
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    p = np.full((Ho * 2, Wo * 2), -np.inf)
    p[:H, :W] = a
    return p.reshape(Ho, 2, Wo, 2).max(axis=(1, 3))


def four_beam(velo, P_velo2im, im_h, im_w, shape=(384, 1280)):
    """kitti_dataset.py:104-106 + mono_dataset.py:196-198: the "4beam" network input [shape/2] (float32, metres / 100)."""
    d = max_pool2x2_ceil(pad_to_shape(depth_image(velo, P_velo2im, im_h, im_w), shape))
    return d.astype(np.float32) / np.float32(100.0)
