"""Oracle restatement of reference ``layers.py`` (fp32, CPU, differentiable via autograd).

Each function cites the reference lines it follows.  Written from the survey's
description of the math, not transcribed: module state (pixel grids etc.) is
rebuilt on the fly so the functions are pure.
"""
import math

import torch
import torch.nn.functional as F

SSIM_C1 = 0.01 ** 2
SSIM_C2 = 0.03 ** 2


def disp_to_depth(disp, min_depth, max_depth):
    """layers.py:11-20 — sigmoid output -> (scaled disparity, depth)."""
    lo = 1 / max_depth
    hi = 1 / min_depth
    scaled = lo + (hi - lo) * disp
    return scaled, 1 / scaled


def rot_from_axisangle(vec):
    """layers.py:59-97 — Rodrigues; ``vec`` is [B,1,3]; returns [B,4,4]."""
    angle = torch.norm(vec, 2, 2, True)                 # [B,1,1]
    axis = vec / (angle + 1e-7)
    ca, sa = torch.cos(angle), torch.sin(angle)
    C = 1 - ca
    x, y, z = (axis[..., i].unsqueeze(1) for i in range(3))
    xs, ys, zs = x * sa, y * sa, z * sa
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    B = vec.shape[0]
    rows = [
        [x * xC + ca, xyC - zs, zxC + ys],
        [xyC + zs, y * yC + ca, yzC - xs],
        [zxC - ys, yzC + xs, z * zC + ca],
    ]
    zero = torch.zeros(B, 1, dtype=vec.dtype)
    out = [torch.cat([e.reshape(B, 1) for e in row] + [zero], 1) for row in rows]
    last = torch.zeros(B, 4, dtype=vec.dtype)
    last[:, 3] = 1
    out.append(last)
    return torch.stack(out, 1)


def get_translation_matrix(t):
    """layers.py:43-56 — [B,1,3] (or [B,3]) -> homogeneous translation [B,4,4]."""
    B = t.shape[0]
    eye = torch.eye(4, dtype=t.dtype).unsqueeze(0).repeat(B, 1, 1)
    tv = torch.cat([t.contiguous().view(B, 3, 1), torch.zeros(B, 1, 1, dtype=t.dtype)], 1)
    return eye + torch.cat([torch.zeros(B, 4, 3, dtype=t.dtype), tv], 2)


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:23-40 — M = T·R, or Rᵀ·T(−t) when ``invert``."""
    R = rot_from_axisangle(axisangle)
    t = translation.clone()
    if invert:
        R = R.transpose(1, 2)
        t = t * -1
    T = get_translation_matrix(t)
    return torch.matmul(R, T) if invert else torch.matmul(T, R)


def pixel_grid(height, width, dtype=torch.float32):
    """layers.py:143-155 — homogeneous pixel coordinates [3, H*W], index = y*W + x."""
    ys, xs = torch.meshgrid(torch.arange(height, dtype=dtype), torch.arange(width, dtype=dtype), indexing="ij")
    return torch.stack([xs.reshape(-1), ys.reshape(-1), torch.ones(height * width, dtype=dtype)], 0)


def backproject_depth(depth, inv_K):
    """layers.py:157-162 — depth[B,1,H,W], inv_K[B,4,4] -> camera points [B,4,H*W]."""
    B, _, H, W = depth.shape
    pix = pixel_grid(H, W, depth.dtype).unsqueeze(0).expand(B, -1, -1)
    rays = torch.matmul(inv_K[:, :3, :3], pix)
    pts = depth.view(B, 1, -1) * rays
    return torch.cat([pts, torch.ones(B, 1, H * W, dtype=depth.dtype)], 1)


def project_3d(points, K, T, height, width, eps=1e-7):
    """layers.py:215-226 — camera points -> normalised sampling grid [B,H,W,2]."""
    B = points.shape[0]
    P = torch.matmul(K, T)[:, :3, :]
    cam = torch.matmul(P, points)
    uv = cam[:, :2, :] / (cam[:, 2, :].unsqueeze(1) + eps)
    uv = uv.view(B, 2, height, width).permute(0, 2, 3, 1)
    u = uv[..., 0] / (width - 1)
    v = uv[..., 1] / (height - 1)
    return (torch.stack([u, v], -1) - 0.5) * 2


def cat_xy(depth, inv_K):
    """layers.py:187-201 (Cat_xy.forward) — normalised xyz map [B,3,H,W] (refiner only)."""
    B, _, H, W = depth.shape
    pix = pixel_grid(H, W, depth.dtype).unsqueeze(0).expand(B, -1, -1)
    pts = depth.view(B, 1, -1) * torch.matmul(inv_K[:, :3, :3], pix)
    pts = pts.view(B, 3, H, W)
    return torch.stack([pts[:, 0] / 30.0, pts[:, 1] / 2.0, (pts[:, 2] - 40) / 40.0], 1)


def upsample(x):
    """layers.py:229-232 — nearest ×2."""
    return F.interpolate(x, scale_factor=2, mode="nearest")


def get_smooth_loss(disp, img):
    """layers.py:235-248 — edge-aware first-order smoothness."""
    ddx = (disp[:, :, :, :-1] - disp[:, :, :, 1:]).abs()
    ddy = (disp[:, :, :-1, :] - disp[:, :, 1:, :]).abs()
    idx = (img[:, :, :, :-1] - img[:, :, :, 1:]).abs().mean(1, keepdim=True)
    idy = (img[:, :, :-1, :] - img[:, :, 1:, :]).abs().mean(1, keepdim=True)
    return (ddx * torch.exp(-idx)).mean() + (ddy * torch.exp(-idy)).mean()


def ssim(x, y):
    """layers.py:267-281 — SSIM *loss* map clamp((1-SSIM)/2, 0, 1), reflect-pad 1, 3x3 box."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect")
    yp = F.pad(y, (1, 1, 1, 1), mode="reflect")
    mu_x = F.avg_pool2d(xp, 3, 1)
    mu_y = F.avg_pool2d(yp, 3, 1)
    sig_x = F.avg_pool2d(xp ** 2, 3, 1) - mu_x ** 2
    sig_y = F.avg_pool2d(yp ** 2, 3, 1) - mu_y ** 2
    sig_xy = F.avg_pool2d(xp * yp, 3, 1) - mu_x * mu_y
    num = (2 * mu_x * mu_y + SSIM_C1) * (2 * sig_xy + SSIM_C2)
    den = (mu_x ** 2 + mu_y ** 2 + SSIM_C1) * (sig_x + sig_y + SSIM_C2)
    return torch.clamp((1 - num / den) / 2, 0, 1)


def conv3x3(x, weight, bias, use_refl=True):
    """layers.py:115-130 — pad(1) (reflect or zero) then 3x3 valid conv with bias."""
    xp = F.pad(x, (1, 1, 1, 1), mode="reflect" if use_refl else "constant")
    return F.conv2d(xp, weight, bias)


def conv_block(x, weight, bias):
    """layers.py:100-112 — Conv3x3 + ELU."""
    return F.elu(conv3x3(x, weight, bias))


def compute_depth_errors(gt, pred):
    """layers.py:284-302 — (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    ratio = torch.max(gt / pred, pred / gt)
    a1 = (ratio < 1.25).float().mean()
    a2 = (ratio < 1.25 ** 2).float().mean()
    a3 = (ratio < 1.25 ** 3).float().mean()
    diff = gt - pred
    rmse = torch.sqrt((diff ** 2).mean())
    rmse_log = torch.sqrt(((torch.log(gt) - torch.log(pred)) ** 2).mean())
    abs_rel = (diff.abs() / gt).mean()
    sq_rel = (diff ** 2 / gt).mean()
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3
