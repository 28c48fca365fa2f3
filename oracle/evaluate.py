"""Oracle restatement of the metric core of reference ``evaluate_depth.py`` (numpy).  TEST INFRASTRUCTURE ONLY.

``compute_errors`` (:42-60) and ``batch_post_process_disparity`` (:62-70) are pinned to goldens produced by the reference's own
functions (tests/golden/make_golden.py::gold_evaluate).  The per-image loop (:344-478) is restated here; its
``cv2.resize`` (OpenCV is not installed in this image and the reference has no vector for it, so this one stage is **parity
unpinned**) is restated from OpenCV's published float32 INTER_LINEAR algorithm: its coefficient rule (double scale, float
coordinate, floor, edge rule) and its two float32 passes, horizontal first (``resize_bilinear``).
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


def _linear_coeffs(n_in, n_out):
    """OpenCV's coefficient rule for INTER_LINEAR (opencv/modules/imgproc/src/resize.cpp, cv::resize -> the `interpolation ==
    INTER_LINEAR` branch that fills xofs / alpha and yofs / beta, OpenCV 3.x - 4.x): scale in double, the source coordinate
    fx = (float)((d + 0.5) * scale - 0.5) rounded to float, s = cvFloor(fx), fx -= s in float; a coordinate left of the first
    sample gives (s, fx) = (0, 0), one at or beyond the last sample (n_in - 1, 0); weights (1.f - fx, fx) as floats."""
    scale = float(n_in) / float(n_out)
    f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo, hi = s < 0, s >= n_in - 1
    s = np.where(lo, 0, np.where(hi, n_in - 1, s))
    f = np.where(lo | hi, np.float32(0), f).astype(np.float32)
    return s, np.minimum(s + 1, n_in - 1), (np.float32(1) - f).astype(np.float32), f


def resize_bilinear(img, out_h, out_w):
    """cv2.resize(img, (out_w, out_h)) (default INTER_LINEAR) on a float32 image, evaluate_depth.py:349.  Restated from OpenCV's
    published algorithm for CV_32F (resize.cpp: HResizeLinear<float, float, float, 1> then VResizeLinear<float, float, float, Cast>):
    coefficients by ``_linear_coeffs``, a HORIZONTAL pass S[sx] * a0 + S[sx + 1] * a1 in float32 per source row, then the vertical
    pass R0 * b0 + R1 * b1 in float32.  **Parity unpinned**: OpenCV is not installed in this image and the reference holds no vector
    for this call; what cannot be restated from the source is whether a given OpenCV build contracts the multiply-adds (its SIMD
    paths use v_muladd, i.e. FMA where the CPU has it) - a <= 1-ulp effect per pixel."""
    img = np.asarray(img, np.float32)
    y0, y1, b0, b1 = _linear_coeffs(img.shape[0], out_h)
    x0, x1, a0, a1 = _linear_coeffs(img.shape[1], out_w)
    rows = (img[:, x0] * a0[None, :] + img[:, x1] * a1[None, :]).astype(np.float32)           # every source row, horizontally resized
    return (rows[y0] * b0[:, None] + rows[y1] * b1[:, None]).astype(np.float32)


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
