"""Oracle restatement of the metric core of reference ``evaluate_depth.py`` (numpy).  TEST INFRASTRUCTURE ONLY.

``compute_errors`` (:42-60) and ``batch_post_process_disparity`` (:62-70) are pinned to goldens produced by the reference's own
functions (tests/golden/make_golden.py::gold_evaluate).  The per-image loop (:344-478) is restated here; its
``cv2.resize`` (OpenCV is not installed in this image, so this one stage is **parity unpinned**) is restated as what
INTER_LINEAR is for float images: half-pixel-centre bilinear interpolation with edge replication.
"""
import numpy as np

MIN_DEPTH = 1e-3
MAX_DEPTH = 80


def compute_errors(gt, pred):
    """evaluate_depth.py:42-60."""
    thresh = np.maximum((gt / pred), (pred / gt))
    a1, a2, a3 = (thresh < 1.25).mean(), (thresh < 1.25 ** 2).mean(), (thresh < 1.25 ** 3).mean()
    rmse = np.sqrt(((gt - pred) ** 2).mean())
    rmse_log = np.sqrt(((np.log(gt) - np.log(pred)) ** 2).mean())
    abs_rel = np.mean(np.abs(gt - pred) / gt)
    sq_rel = np.mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


def batch_post_process_disparity(l_disp, r_disp):
    """evaluate_depth.py:62-70."""
    _, h, w = l_disp.shape
    m_disp = 0.5 * (l_disp + r_disp)
    l, _ = np.meshgrid(np.linspace(0, 1, w), np.linspace(0, 1, h))
    l_mask = (1.0 - np.clip(20 * (l - 0.05), 0, 1))[None, ...]
    r_mask = l_mask[:, :, ::-1]
    return r_mask * l_disp + l_mask * r_disp + (1.0 - l_mask - r_mask) * m_disp


def resize_bilinear(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) with INTER_LINEAR on a float32 image."""
    h, w = img.shape
    ys = np.clip((np.arange(out_h) + 0.5) * (h / out_h) - 0.5, 0, h - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * (w / out_w) - 0.5, 0, w - 1)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (ys - y0).astype(np.float32)[:, None], (xs - x0).astype(np.float32)[None, :]
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return (top * (1 - fy) + bot * fy).astype(np.float32)


def evaluate_predictions(pred_disps, gt_depths, eval_split="eigen", pred_depth_scale_factor=1.0, disable_median_scaling=False):
    """evaluate_depth.py:344-478 -> (mean errors [7], ratios)."""
    errors, ratios = [], []
    for i in range(len(gt_depths)):
        gt_depth = gt_depths[i]
        gh, gw = gt_depth.shape[:2]
        pred_depth = 1 / resize_bilinear(pred_disps[i], gh, gw)
        if eval_split in ("eigen", "demo"):
            mask = np.logical_and(gt_depth > MIN_DEPTH, gt_depth < MAX_DEPTH)
            crop = np.array([0.40810811 * gh, 0.99189189 * gh, 0.03594771 * gw, 0.96405229 * gw]).astype(np.int32)
            crop_mask = np.zeros(mask.shape)
            crop_mask[crop[0]:crop[1], crop[2]:crop[3]] = 1
            mask = np.logical_and(mask, crop_mask)
        else:
            mask = gt_depth > 0
        pred_depth *= pred_depth_scale_factor
        if not disable_median_scaling:
            ratio = np.median(gt_depth[mask]) / np.median(pred_depth[mask])
            ratios.append(ratio)
            pred_depth *= ratio
        pred_depth, gt_m = pred_depth[mask], gt_depth[mask]
        pred_depth[pred_depth < MIN_DEPTH] = MIN_DEPTH
        pred_depth[pred_depth > MAX_DEPTH] = MAX_DEPTH
        errors.append(compute_errors(gt_m, pred_depth))
    return np.array(errors).mean(0), np.array(ratios)
