"""Device-side mirror of the metric core of the reference ``evaluate_depth.py``.

  * ``compute_errors(gt, pred)``                      evaluate_depth.py:42-60   (fd_depth_errors)
  * ``batch_post_process_disparity(l_disp, r_disp)``  evaluate_depth.py:62-70   (fd_post_process_disparity, float64 like numpy)
  * ``evaluate_predictions(pred_disps, gt_depths, ...)``  the per-image loop of ``evaluate`` (evaluate_depth.py:344-478):
    resize the predicted disparity to the ground-truth size, invert, Eigen mask + Garg crop, ``pred_depth_scale_factor``,
    median scaling, clamp to [1e-3, 80], metrics, mean over images.
Model loading, the dataset walk, colour-mapped PNG dumps and the per-semantic-class breakdown of the reference script are
outside the hot path.  No CPU fallback: tensors must live on the GPU.
"""
import numpy as np
import torch

from . import functional as FD
from ._lib import call, stream

MIN_DEPTH = 1e-3          # evaluate_depth.py:28-29
MAX_DEPTH = 80


def compute_errors(gt, pred):
    """evaluate_depth.py:42-60 on matched device tensors -> (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3) as floats."""
    return tuple(float(v) for v in FD.depth_errors(gt, pred))


def batch_post_process_disparity(l_disp, r_disp):
    """evaluate_depth.py:62-70: [B,H,W] float32 device tensors -> [B,H,W] float64 device tensor."""
    l_disp, r_disp = FD.f32(l_disp), FD.f32(r_disp)
    FD._need_cuda(l_disp, r_disp)
    assert l_disp.shape == r_disp.shape and l_disp.dim() == 3
    out = torch.empty(l_disp.shape, device=l_disp.device, dtype=torch.float64)
    call("fd_post_process_disparity", l_disp.data_ptr(), r_disp.data_ptr(), out.data_ptr(), l_disp.shape[0], l_disp.shape[1],
         l_disp.shape[2], stream())
    return out


def _median(v):
    """np.median (mean of the two middle values for an even count; torch.median would take the lower one)."""
    s, _ = torch.sort(v.reshape(-1))
    n = s.numel()
    return s[n // 2] if n % 2 else 0.5 * (s[n // 2 - 1] + s[n // 2])


def garg_crop(gt_height, gt_width):
    """evaluate_depth.py:361-363."""
    return np.array([0.40810811 * gt_height, 0.99189189 * gt_height, 0.03594771 * gt_width, 0.96405229 * gt_width]).astype(np.int32)


def evaluate_predictions(pred_disps, gt_depths, eval_split="eigen", pred_depth_scale_factor=1.0, disable_median_scaling=False):
    """evaluate_depth.py:344-478.  ``pred_disps``: [N,h,w] device tensor (or list); ``gt_depths``: list of [H_i,W_i] arrays /
    tensors (KITTI ground truth has per-drive sizes).  Returns (mean of the 7 metrics over the images, per-image scaling ratios)."""
    errors, ratios = [], []
    for i in range(len(gt_depths)):
        gt = torch.as_tensor(gt_depths[i], dtype=torch.float32).cuda()
        gh, gw = gt.shape
        disp = FD.f32(torch.as_tensor(pred_disps[i])).cuda()[None, None]
        # cv2.resize(pred_disp, (gt_width, gt_height)): OpenCV's float32 INTER_LINEAR rule on the device (fd_resize_linear_cv; parity
        # unpinned - no OpenCV in the build image - and restated from its source, like oracle/evaluate.py::resize_bilinear)
        disp = FD.resize_linear_cv(disp, (gh, gw))
        pred_depth = 1.0 / disp[0, 0]
        if eval_split in ("eigen", "demo"):
            mask = (gt > MIN_DEPTH) & (gt < MAX_DEPTH)
            c = garg_crop(gh, gw)
            crop = torch.zeros_like(mask)
            crop[c[0]:c[1], c[2]:c[3]] = True
            mask = mask & crop
        else:
            mask = gt > 0
        pred_depth = pred_depth * pred_depth_scale_factor
        if not disable_median_scaling:
            ratio = _median(gt[mask]) / _median(pred_depth[mask])
            ratios.append(float(ratio))
            pred_depth = pred_depth * ratio
        pred, g = torch.clamp(pred_depth[mask], MIN_DEPTH, MAX_DEPTH), gt[mask]
        errors.append(compute_errors(g, pred))
    return np.array(errors).mean(0), np.array(ratios)
