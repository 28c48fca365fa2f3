"""Deterministic synthetic inputs shared by ``make_golden.py`` (which feeds them to the imported
reference) and by the tests (which feed them to the oracle / the HIP path).  numpy MT19937 streams
are stable across numpy versions, so inputs are regenerated from a seed instead of being stored.
"""
import numpy as np
import torch

K_NORM = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)


def _box3(a):
    p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(1, 1), (1, 1)], mode="edge")
    out = np.zeros_like(a)
    H, W = a.shape[-2:]
    for dy in range(3):
        for dx in range(3):
            out += p[..., dy:dy + H, dx:dx + W]
    return out / 9.0


def smooth_image(rng, B, C, H, W):
    a = rng.rand(B, C, H, W).astype(np.float32)
    return _box3(_box3(a)).astype(np.float32)


def intrinsics(B, H, W, scale):
    """kitti_dataset.py:36-39 + mono_dataset.py:166-175 — K scaled to (W,H)/2^scale, inv via pinv."""
    K = K_NORM.copy()
    K[0, :] *= W // (2 ** scale)
    K[1, :] *= H // (2 ** scale)
    inv_K = np.linalg.pinv(K)
    return (torch.from_numpy(np.repeat(K[None], B, 0).copy()),
            torch.from_numpy(np.repeat(inv_K[None], B, 0).astype(np.float32).copy()))


def lidar_4beam(rng, B, H, W, density=3):
    """4 scan rows in the lower half, every ``density``-th column, depth U[5,65] m, stored /100."""
    beam = np.zeros((B, 1, H, W), dtype=np.float32)
    rows = [int(H * f) for f in (0.52, 0.625, 0.73, 0.835)]
    for b in range(B):
        for r in rows:
            cols = np.arange(2 + (b + r) % density, W - 2, density)
            beam[b, 0, r, cols] = rng.uniform(5.0, 65.0, size=cols.shape).astype(np.float32) / 100.0
    return beam


def lidar_random(rng, B, H, W, n_points, roi=(76, 190, 2, 638), lo=5.0, hi=65.0):
    """The r100 / r200 inputs of BASELINE config 5: ``n_points`` uniformly random pixels of the ROI per image
    carry a return, depth U[lo, hi] m, stored / 100 like the 4-beam maps (the reference's ``random100`` / ``random200`` scans)."""
    beam = np.zeros((B, 1, H, W), dtype=np.float32)
    r0, r1, c0, c1 = roi
    r0, r1, c0, c1 = min(r0, H - 1), min(r1, H), min(c0, W - 1), min(c1, W)
    for b in range(B):
        flat = rng.choice((r1 - r0) * (c1 - c0), size=n_points, replace=False)
        rows, cols = r0 + flat // (c1 - c0), c0 + flat % (c1 - c0)
        beam[b, 0, rows, cols] = rng.uniform(lo, hi, size=n_points).astype(np.float32) / 100.0
    return beam


def batch_inputs(seed, B, H, W, num_scales=4, frame_ids=(0, -1, 1)):
    """Dict with the schema of mono_dataset.py:109-228 (tensor entries only)."""
    rng = np.random.RandomState(seed)
    base = smooth_image(rng, B, 3, H, W + 8)
    inputs = {}
    for f in frame_ids:
        shift = 4 + 2 * f
        img = base[..., shift:shift + W] + 0.01 * rng.randn(B, 3, H, W).astype(np.float32)
        img = np.clip(img, 0, 1).astype(np.float32)
        for s in range(num_scales):
            t = torch.from_numpy(img.copy())
            if s > 0:
                t = torch.nn.functional.avg_pool2d(t, 2 ** s)
            inputs[("color", f, s)] = t
            inputs[("color_aug", f, s)] = t.clone()
    for s in range(num_scales):
        inputs[("K", s)], inputs[("inv_K", s)] = intrinsics(B, H, W, s)
    inputs["4beam"] = torch.from_numpy(lidar_4beam(rng, B, H, W))
    return inputs, rng


def disp_pyramid(rng, B, H, W, num_scales=4):
    """Sigmoid-range disparity maps chosen so depth*26 overlaps the LiDAR range (SI-loss mask non-empty)."""
    out = {}
    for s in range(num_scales):
        h, w = H // 2 ** s, W // 2 ** s
        d = smooth_image(rng, B, 1, h, w)
        out[("disp", s)] = torch.from_numpy((0.03 + 0.10 * d).astype(np.float32))
    return out


def make_si_mask_empty(inputs, disp, mode):
    """In place: empty the LiDAR validity mask (trainer.py:580-584) - ``"all"``: no LiDAR returns; ``"scale2"``: the 1/4-scale
    disparity is 0.9 everywhere (26 / (0.01 + 9.99 * 0.9) = 2.9 m, further than 2 m from every 5 .. 65 m return); ``"sample0"``:
    the first sample of the batch has no returns (an empty micro-batch inside a stacked pass); ``None``: nothing."""
    if mode == "all":
        inputs["4beam"].zero_()
    elif mode == "scale2":
        disp[("disp", 2)].fill_(0.9)
    elif mode == "sample0":
        inputs["4beam"][0].zero_()
    elif mode is not None:
        raise ValueError(mode)


def small_poses(rng, B):
    """(axisangle[B,1,3], translation[B,1,3]) of KITTI-like magnitude."""
    aa = (0.02 * rng.randn(B, 1, 3)).astype(np.float32)
    tr = (0.05 * rng.randn(B, 1, 3)).astype(np.float32)
    tr[..., 2] += 0.1
    return torch.from_numpy(aa), torch.from_numpy(tr)


def feature_pyramids(rng, B, H, W, ch):
    """Two ReLU-like 5-level feature pyramids (RGB encoder / beam encoder stand-ins) for an HxW input."""
    feats, beams = [], []
    for i, c in enumerate(ch):
        h, w = H // 2 ** (i + 1), W // 2 ** (i + 1)
        feats.append(torch.from_numpy(np.maximum(rng.randn(B, c, h, w), 0).astype(np.float32)))
        beams.append(torch.from_numpy(np.maximum(rng.randn(B, c, h, w), 0).astype(np.float32)))
    return feats, beams


def fill_params(module, seed):
    """Overwrite every parameter/buffer of ``module`` (state-dict order) with seeded numpy values:
    weights ~ N(0, 2/fan_in), biases ~ N(0, 0.01), BN weight ~ U[0.5,1.5], running_var ~ U[0.5,1.5]."""
    rng = np.random.RandomState(seed)
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if name.endswith("num_batches_tracked"):
                continue
            shape = tuple(t.shape)
            if t.dim() >= 2:
                fan_in = int(np.prod(shape[1:]))
                v = rng.randn(*shape) * np.sqrt(2.0 / fan_in)
            elif name.endswith("running_var") or (name.endswith("weight") and t.dim() == 1):
                v = rng.uniform(0.5, 1.5, size=shape)
            elif name.endswith("running_mean"):
                v = rng.randn(*shape) * 0.1
            else:
                v = rng.randn(*shape) * 0.01
            t.copy_(torch.from_numpy(v.astype(np.float32)))
    return module


def depth_eval_inputs(seed, B, H, W, gt_h=375, gt_w=1242):
    """Sparse ground-truth depth [B,1,375,1242] (about 5 % valid, metres) and a dense prediction [B,1,H,W] for the
    monitoring metrics of trainer.py:598-630."""
    rng = np.random.RandomState(seed)
    gt = rng.uniform(1.5, 70.0, size=(B, 1, gt_h, gt_w)).astype(np.float32)
    gt[rng.rand(B, 1, gt_h, gt_w) > 0.05] = 0.0
    pred = rng.uniform(0.5, 90.0, size=(B, 1, H, W)).astype(np.float32)
    return gt, pred


def refiner_inputs(seed, B, H, W):
    """Inputs of one refiner step: the trainer batch + a dense "GDC" depth map (metres) + LiDAR maps made by the oracle's
    scatter (shared by the generator and the tests)."""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from oracle import scatter as OS
    inp, rng = batch_inputs(seed, B, H, W)
    roi = (76, 190, 2, 638)
    for i, f in enumerate((0, -1, 1)):
        beam = lidar_4beam(np.random.RandomState(seed + 10 + i), B, H, W)
        beam = np.where(beam > 0, 0.035 + (beam - 0.05) * (0.035 / 0.6), 0).astype(np.float32)     # ~3.5-7 m returns
        two = np.stack([np.stack(OS.scatter_2channel_c(beam[b, 0], roi)) for b in range(B)])
        inp[("2channel", f, 0)] = torch.from_numpy(two)
        if f == 0:
            inp["2channel"] = torch.from_numpy(two)
            inp["4beam"] = torch.from_numpy(beam)
    # dense "GDC" depth in the units of the refined prediction (disp ~ 0.5 -> depth ~ 0.2), so that the SI-log mask
    # |pred - target| < gdc_loss_threshold (refiner.py:561-562) is well populated
    inp["inf_gdc"] = torch.from_numpy(np.random.RandomState(seed + 77).uniform(0.05, 1.5, size=(B, 1, H, W)).astype(np.float32))
    noise = [torch.from_numpy(np.random.RandomState(seed + 50 + s).randn(B, 2, H, W).astype(np.float32)) for s in range(4)]
    return inp, noise


def refiner_models(module_table, seed=5):
    """The seven networks of refiner.py:80-160 with seeded weights (tests rebuild the same state from the seeds).  The
    refine decoder's four disparity heads are scaled down so that sigmoid(.) stays away from saturation: with raw random
    weights the refined disparity spans [1e-8, 1] (depth 0.1 .. 100 m), almost every warp lands on the image border and the
    gradient is carried by a few hundred pixels whose argmin / clamp branches flip with the last bit of rounding."""
    for k, net in module_table.items():
        fill_params(net, 300 + len(k) + seed)
        net.train() if k == "refine2d_decoder" else net.eval()
    with torch.no_grad():
        for name, t in module_table["refine2d_decoder"].state_dict().items():
            if name.split(".")[1] in ("10", "11", "12", "13"):
                t.mul_(0.02)
    return module_table


def lidar_scan(seed, n_points=6000, im_h=375, im_w=1242):
    """A synthetic Velodyne scan (float32 [N,4]: forward, left, up, reflectance) and a KITTI-like velodyne -> image
    projection (float64 [3,4]).  Includes points behind the camera, points outside the image, clusters that pile up on
    single pixels and points forced onto the first / last image column of adjacent rows (the reference's sub2ind quirk)."""
    rng = np.random.RandomState(seed)
    fwd = rng.uniform(-5.0, 70.0, n_points)
    left = rng.uniform(-25.0, 25.0, n_points)
    up = rng.uniform(-2.5, 1.5, n_points)
    velo = np.stack([fwd, left, up, rng.rand(n_points)], 1).astype(np.float32)
    velo[:400, :3] = velo[400:800, :3] * np.float32(1.0003)          # near-duplicates: many pixels get 2+ points
    K = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]])
    R = np.eye(4)
    c, s_ = np.cos(0.01), np.sin(0.01)
    R[:3, :3] = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    velo2cam = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                         [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                         [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01], [0.0, 0.0, 0.0, 1.0]])
    P = np.dot(np.dot(K, R), velo2cam)                       # kitti_utils.py:57, same association
    # points whose projection lands exactly on column 0 / column W-1 of adjacent rows
    Pi = np.linalg.pinv(np.vstack([P, [0, 0, 0, 1.0]]))
    extra = []
    for r in (100, 101, 102, 200):
        for col, zz in ((0, 12.0), (im_w - 1, 9.0), (0, 7.5)):
            uvw = np.array([(col + 1.0) * zz, (r + 1.0 - (col != 0)) * zz, zz, 1.0])
            x = Pi @ uvw
            extra.append([x[0] / x[3], x[1] / x[3], x[2] / x[3], 0.5])
    velo = np.concatenate([velo, np.array(extra, dtype=np.float32)], 0)
    lidar_scan.calib = dict(P_rect_02=K, R_rect_00=R[:3, :3].copy(), R=velo2cam[:3, :3].copy(), T=velo2cam[:3, 3].copy(),
                            S_rect_02=np.array([float(im_w), float(im_h)]))
    return velo, P



def eval_pairs(seed):
    """Matched (ground truth, prediction) depth vectors in metres for evaluate_depth.compute_errors: float32, 1-D."""
    rng = np.random.RandomState(seed)
    out = []
    for n, noise in ((5000, 0.05), (20011, 0.3), (37, 1.0)):
        gt = rng.uniform(1.0, 80.0, n).astype(np.float32)
        pred = np.clip(gt * np.exp(rng.randn(n) * noise), 1e-3, 80).astype(np.float32)
        out.append((gt, pred))
    return out


def disp_pair(seed, B, H, W):
    """Disparities of an image batch and of its mirrored twin (already flipped back), float32 [B,H,W]."""
    rng = np.random.RandomState(seed)
    l = rng.uniform(0.01, 0.9, (B, H, W)).astype(np.float32)
    r = (l * rng.uniform(0.8, 1.2, (B, H, W))).astype(np.float32)
    return l, r
