"""Synthetic KITTI-shaped minibatches with the schema of the reference data loader
(datasets/mono_dataset.py:109-228 as consumed by trainer.py:268-319), generated on the device.

There is no KITTI (and no network) in the build/bench environment, so the benchmark and the parity tests
feed tensors of the right shapes and statistics instead: low-pass colour images whose frames -1/+1 are
shifted copies of frame 0 (so the photometric loss has signal), KITTI's normalised intrinsics, a 4-row
"4-beam" LiDAR map and its 2-channel expansion computed by the scatter kernel itself.
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
