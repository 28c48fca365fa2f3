"""Drop-in for the reference ``layers.py`` (same names, arguments and semantics), HIP-backed.

Reference lines are cited per symbol.  Modules keep the reference's constructor signatures
(e.g. ``BackprojectDepth(batch_size, height, width)``) although the kernels need no baked-in
pixel-grid buffers: coordinates are generated in registers.
"""
import torch
import torch.nn as nn

from . import functional as FD

disp_to_depth = FD.disp_to_depth                                    # layers.py:11-20
transformation_from_parameters = FD.transformation_from_parameters  # layers.py:23-40
get_smooth_loss = FD.get_smooth_loss                                # layers.py:235-248


def rot_from_axisangle(vec):
    """layers.py:59-97 — [B,1,3] -> [B,4,4] rotation (translation column zero)."""
    return FD.transformation_from_parameters(vec, torch.zeros_like(vec), invert=False)


def get_translation_matrix(translation_vector):
    """layers.py:43-56 — translation -> [B,4,4]."""
    return FD.transformation_from_parameters(torch.zeros_like(translation_vector), translation_vector, invert=False)


class Conv3x3(nn.Module):
    """layers.py:115-130 — ReflectionPad2d(1) (or zero pad) + 3x3 conv with bias.
    State-dict keys: ``conv.weight``, ``conv.bias`` (as the reference's nn.Conv2d child)."""

    def __init__(self, in_channels, out_channels, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.conv = nn.Conv2d(int(in_channels), int(out_channels), 3)   # parameter holder only

    def forward(self, x):
        return FD.conv2d(x, self.conv.weight, self.conv.bias, stride=1, pad=1,
                         pad_mode="reflect" if self.use_refl else "zero", act="none")


class ConvBlock(nn.Module):
    """layers.py:100-112 — Conv3x3 + ELU (fused into the conv epilogue)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = Conv3x3(in_channels, out_channels)

    def forward(self, x, grad_preact=False, channels=None):
        """``grad_preact``: the caller guarantees that the ONE consumer of the result returns the gradient w.r.t. this block's
        pre-activation (FD.conv2d: the ELU' pass then runs inside that consumer's backward kernel).
        ``channels`` = (Cin_p, Cout_p): run the block on channel-padded operands - ``x`` carries Cin_p >= Cin channels (the extra ones
        zero), the result Cout_p >= Cout (the extra ones ELU(0) = 0): the parameters are zero-padded per call (autograd crops their
        gradients), which puts layers whose channel counts are not multiples of 16 - the refine decoder's 262 / 134 / 102 / 22 - on
        the MFMA fast-path / Winograd kernels instead of the generic gather GEMM (networks/depth_decoder.py)."""
        c = self.conv.conv
        w, b = c.weight, c.bias
        if channels is not None and (channels[0] != w.shape[1] or channels[1] != w.shape[0]):
            w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, channels[0] - w.shape[1], 0, channels[1] - w.shape[0]))
            b = torch.nn.functional.pad(b, (0, channels[1] - b.shape[0]))
        return FD.conv2d(x, w, b, stride=1, pad=1, pad_mode="reflect", act="elu", grad_preact=grad_preact)


class BackprojectDepth(nn.Module):
    """layers.py:133-162."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        return FD.backproject_depth(depth.reshape(-1, 1, self.height, self.width), inv_K)


class Project3D(nn.Module):
    """layers.py:204-226."""

    def __init__(self, batch_size, height, width, eps=1e-7):
        super().__init__()
        self.batch_size, self.height, self.width, self.eps = batch_size, height, width, eps

    def forward(self, points, K, T):
        return FD.project_3d(points, K, T, self.height, self.width, self.eps)


class Cat_xy(nn.Module):
    """layers.py:165-201 (refiner input channels)."""

    def __init__(self, batch_size, height, width):
        super().__init__()
        self.batch_size, self.height, self.width = batch_size, height, width

    def forward(self, depth, inv_K):
        return FD.cat_xy(depth.reshape(-1, 1, self.height, self.width), inv_K)


class SSIM(nn.Module):
    """layers.py:251-281."""

    def forward(self, x, y):
        return FD.ssim(x, y)


def upsample(x):
    """layers.py:229-232 — nearest x2."""
    return FD.upsample_nearest2x(x)


def compute_depth_errors(gt, pred):
    """layers.py:284-302 — seven depth metrics over already-masked vectors."""
    return FD.depth_errors(gt, pred)
