"""Synthetic KITTI-shaped minibatches with the schema of the reference data loader
(datasets/mono_dataset.py:109-228 as consumed by trainer.py:268-319), generated on the device.

There is no KITTI (and no network) in the build/bench environment, so the benchmark and the parity tests
feed tensors of the right shapes and statistics instead.  Two generators:
  * ``make_batch``: low-pass colour images whose frames -1/+1 are shifted copies of frame 0, KITTI's normalised
    intrinsics, a 4-row "4-beam" LiDAR map with random ranges and its 2-channel expansion (scatter kernel);
  * ``make_scene_batch``: a CONSISTENT scene - a ground-truth depth field, frames -1/+1 rendered from frame 0 through
    that depth and a ground-truth ego-motion with the loss path's own projection convention, LiDAR returns sampled
    from the depth field, ``depth_gt`` = the field at KITTI's ground-truth size.  Training on it makes progress, the
    LiDAR term keeps valid returns (what trainer.py:577-589 needs to stay defined), and AbsRel against ``depth_gt``
    means something: this is what bench.py feeds and what the AbsRel parity test trains on.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import functional as FD

K_NORMALISED = np.array([[0.58, 0, 0.5, 0], [0, 1.92, 0.5, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)  # kitti_dataset.py:36-39


def intrinsics(batch, height, width, num_scales, device):
    """mono_dataset.py:166-175 — K scaled per pyramid level, inv_K = pinv(K)."""
    out = {}
    for s in range(num_scales):
        K = K_NORMALISED.copy()
        K[0, :] *= width // (2 ** s)
        K[1, :] *= height // (2 ** s)
        inv_K = np.linalg.pinv(K).astype(np.float32)
        out[("K", s)] = torch.from_numpy(K).to(device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
        out[("inv_K", s)] = torch.from_numpy(inv_K).to(device).unsqueeze(0).repeat(batch, 1, 1).contiguous()
    return out


def _smooth(x):
    """Two 3x3 box blurs with replicated borders (plain pooling: input synthesis must not pull a convolution library into the
    process that is being profiled)."""
    for _ in range(2):
        x = F.avg_pool2d(F.pad(x, (1, 1, 1, 1), mode="replicate"), 3, stride=1)
    return x


def lidar_4beam(batch, height, width, gen, device, shift=0):
    """Four scan rows, every 3rd column, depth(m)/100 (kitti_dataset.py:93-117 stores metres/100)."""
    beam = torch.zeros(batch, 1, height, width, device=device)
    rows = [int(height * f) for f in (0.52, 0.625, 0.73, 0.835)]
    for r in rows:
        cols = torch.arange(2 + (r + shift) % 3, width - 2, 3, device=device)
        depth = torch.empty(batch, cols.numel(), device=device).uniform_(3.5, 65.0, generator=gen)
        # every other return lies near what a randomly initialised DepthDecoder predicts (sigmoid(0) -> ~5 m at the SI
        # loss's scale), so the masked SI-log loss (trainer.py:577-589) always has valid points: with none it is NaN by
        # construction (mean / variance of an empty set), in the reference as well
        near = torch.empty(batch, cols.numel(), device=device).uniform_(4.0, 7.0, generator=gen)
        depth[:, ::2] = near[:, ::2]
        beam[:, 0, r, cols] = depth / 100.0
    return beam


def make_batch(batch, height=192, width=640, num_scales=4, frame_ids=(0, -1, 1), seed=1234, device="cuda",
               with_depth_gt=False):
    """One minibatch dict keyed like the reference's (tuple keys), all float32 on ``device``."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    inputs = {}
    base = _smooth(torch.rand(batch, 3, height, width + 8, device=device, generator=gen))
    lo, hi = base.amin(), base.amax()
    base = (base - lo) / (hi - lo)
    for f in frame_ids:
        off = 7 if f == "s" else 4 + 2 * f        # the stereo partner: a 3-pixel horizontal shift
        img = base[..., off:off + width] + 0.01 * torch.randn(batch, 3, height, width, device=device, generator=gen)
        img = img.clamp(0, 1).contiguous()
        for s in range(num_scales):
            lvl = img if s == 0 else F.avg_pool2d(img, 2 ** s)
            inputs[("color", f, s)] = lvl.contiguous()
            inputs[("color_aug", f, s)] = inputs[("color", f, s)]     # no colour augmentation in the synthetic feed
    inputs.update(intrinsics(batch, height, width, num_scales, device))
    roi = FD.scaled_roi(height, width)
    for i, f in enumerate(frame_ids):
        beam = lidar_4beam(batch, height, width, gen, device, shift=i)
        two = FD.scatter_2channel(beam, roi)
        inputs[("2channel", f, 0)] = two
        if f == 0:
            inputs["4beam"] = beam
            inputs["2channel"] = two
    if "s" in frame_ids:
        # datasets/mono_dataset.py:216-222: a pure sideways translation of 0.1 (= the 54 cm baseline in the dataset's units)
        T = torch.eye(4, device=device).repeat(batch, 1, 1)
        T[:, 0, 3] = -0.1
        inputs["stereo_T"] = T
    if with_depth_gt:
        inputs["depth_gt"] = torch.empty(batch, 1, 375, 1242, device=device).uniform_(1.0, 80.0, generator=gen)
    return inputs


# ------------------------------------------------------------------------------------------------ consistent scene
def scene_depth(batch, height, width, gen, device):
    """Ground-truth depth field in metres, [B,1,H,W]: a road-like ramp (far at the top, ~5.5 m at the bottom rows) plus smooth
    bumps; 4.5 .. 34 m.  The LiDAR rows (0.52 .. 0.835 of the height) see 6 .. 20 m."""
    ys = torch.linspace(0.0, 1.0, height, device=device).view(1, 1, height, 1)
    ramp = 5.5 + 26.0 * (1.0 - ys).pow(1.6)
    coarse = torch.rand(batch, 1, height // 16 + 2, width // 16 + 2, device=device, generator=gen)
    bumps = F.interpolate(coarse, size=(height, width), mode="bilinear", align_corners=False) - 0.5
    return (ramp * (1.0 + 0.25 * bumps)).clamp(4.5, 34.0).contiguous()


def scene_motion(batch, frame_id, gen, device):
    """Ground-truth T (frame 0 -> frame ``frame_id``) in the networks' units (metres / 26, the LiDAR term's depth scale -
    options.py:242-249): ~0.5 m of forward motion per frame with a little sideways drift, no rotation."""
    sign = -1.0 if frame_id < 0 else 1.0
    T = torch.eye(4, device=device).repeat(batch, 1, 1)
    jitter = torch.empty(batch, 2, device=device).uniform_(0.8, 1.2, generator=gen)
    T[:, 2, 3] = -sign * 0.019 * jitter[:, 0]
    T[:, 0, 3] = sign * 0.004 * jitter[:, 1]
    return T


def render_frame(image0, depth_units, K, inv_K, T_0_to_f, iterations=5):
    """Frame f of the scene from frame 0 = the exact inverse of the loss path's warp (layers.py:133-162 / 204-226,
    trainer.py:467-470): I_f(q) = I_0(p) where p is the pixel of frame 0 whose scene point (depth D_0(p)) projects to q under
    T_0->f.  p is found by fixed-point iteration p <- p - (project(p) - q), which converges for the smooth depth fields and the
    small motions used here (the flow's gradient is << 1).  Plain torch ops: input synthesis only, the product path never calls this."""
    B, _, H, W = image0.shape
    dev = image0.device
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32),
                            indexing="ij")
    q = torch.stack([xs, ys], 0).unsqueeze(0).expand(B, 2, H, W)
    P = (K @ T_0_to_f)[:, :3, :]
    Kinv = inv_K[:, :3, :3]

    def sample(img, p):         # bilinear lookup at pixel coordinates p [B,2,H,W], borders clamped
        g = torch.stack([(p[:, 0] + 0.5) / W * 2 - 1, (p[:, 1] + 0.5) / H * 2 - 1], -1)
        return F.grid_sample(img, g, padding_mode="border", align_corners=False)

    def project(p):             # where frame 0's pixel p lands in frame f
        d = sample(depth_units, p)
        pix = torch.cat([p, torch.ones(B, 1, H, W, device=dev)], 1).reshape(B, 3, H * W)
        cam = d.reshape(B, 1, H * W) * (Kinv @ pix)
        cam = torch.cat([cam, torch.ones(B, 1, H * W, device=dev)], 1)
        c = P @ cam
        return (c[:, :2] / (c[:, 2:3] + 1e-7)).reshape(B, 2, H, W)

    p = q.clone()
    for _ in range(iterations):
        p = p - (project(p) - q)
    return sample(image0, p)


def make_scene_batch(batch, height=192, width=640, num_scales=4, frame_ids=(0, -1, 1), seed=1234, device="cuda", scatter=None,
                     gt_size=(375, 1242), clutter=0.0):
    """One minibatch of the consistent scene (see the module docstring), reference schema, float32 on ``device``.
    ``device="cpu"`` (tests / fixtures: the same tensors can be fed to the CPU oracle and uploaded for the HIP path) needs
    ``scatter``: a callable beam[B,1,H,W] -> [B,2,H,W] (the oracle's gen2channel restatement); on the GPU the scatter kernel
    is used.  The ground truth rides in the batch as ``depth_gt`` (metres, KITTI's 375x1242) and ``("T_gt", f)``.
    ``clutter``: fraction of the LiDAR returns replaced by clutter spread evenly over 2 .. 78 m (real scans carry outliers too).
    bench.py uses 0.5: whatever a from-scratch network predicts while it is still wandering, some returns then lie within the
    2 m validity window of trainer.py:580-584, so the LiDAR term stays defined (with no valid return it is NaN by construction,
    in the reference as well - tests/test_gpu_losspath.py::test_empty_lidar_mask_*)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    inputs = {}
    depth_m = scene_depth(batch, height, width, gen, device)
    # texture with structure at ~8 px and ~3 px: survives the two bilinear resamplings (rendering, then the loss's warp)
    coarse = F.interpolate(torch.rand(batch, 3, height // 4, width // 4, device=device, generator=gen), size=(height, width),
                           mode="bilinear", align_corners=False)
    base = _smooth(0.75 * coarse + 0.25 * _smooth(torch.rand(batch, 3, height, width, device=device, generator=gen)))
    lo, hi = base.amin(), base.amax()
    image0 = ((base - lo) / (hi - lo)).contiguous()
    inputs.update(intrinsics(batch, height, width, num_scales, device))
    for f in frame_ids:
        if f == 0:
            img = image0
        else:
            T = scene_motion(batch, f, gen, device)
            inputs[("T_gt", f)] = T
            img = render_frame(image0, depth_m / 26.0, inputs[("K", 0)], inputs[("inv_K", 0)], T)
        img = (img + 0.01 * torch.randn(batch, 3, height, width, device=device, generator=gen)).clamp(0, 1).contiguous()
        for s in range(num_scales):
            lvl = img if s == 0 else F.avg_pool2d(img, 2 ** s)
            inputs[("color", f, s)] = lvl.contiguous()
            inputs[("color_aug", f, s)] = inputs[("color", f, s)]
    roi = FD.scaled_roi(height, width)
    rows = [int(height * q) for q in (0.52, 0.625, 0.73, 0.835)]
    for i, f in enumerate(frame_ids):
        beam = torch.zeros(batch, 1, height, width, device=device)
        for r in rows:
            cols = torch.arange(2 + (r + i) % 3, width - 2, 3, device=device)
            rng = 1.0 + 0.01 * torch.randn(batch, cols.numel(), device=device, generator=gen)       # 1 % range noise
            ranges = depth_m[:, 0, r][:, cols] * rng
            if clutter > 0:
                n_cl = int(cols.numel() * clutter)
                pick = torch.randperm(cols.numel(), device=device, generator=gen)[:n_cl]
                ladder = 2.0 + 76.0 * (torch.arange(n_cl, device=device, dtype=torch.float32) + 0.5) / max(n_cl, 1)
                ranges[:, pick] = ladder.unsqueeze(0).expand(batch, n_cl)
            beam[:, 0, r, cols] = ranges / 100.0                                                     # metres / 100
        two = FD.scatter_2channel(beam, roi) if scatter is None else scatter(beam)
        inputs[("2channel", f, 0)] = two
        if f == 0:
            inputs["4beam"], inputs["2channel"] = beam, two
    inputs["depth_gt"] = F.interpolate(depth_m, size=gt_size, mode="bilinear", align_corners=False).contiguous()
    return inputs
