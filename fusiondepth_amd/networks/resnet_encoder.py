"""HIP-backed ``ResnetEncoder`` (reference networks/resnet_encoder.py:53-103).

The reference builds its trunk from torchvision 0.9 (``models.resnet18/34/50/101/152``, or the
multi-image subclass at :11-50).  torchvision is not a dependency here: ``ResNetTrunk`` declares the same
module tree (so ``state_dict()`` keys/shapes are identical: ``encoder.conv1.weight``,
``encoder.layer1.0.bn1.running_mean``, ..., including the unused ``encoder.fc.*``) with ``nn.Conv2d`` /
``nn.BatchNorm2d`` used purely as parameter/buffer holders.  Every forward op is a libfdhip kernel:
MFMA implicit-GEMM convs, BatchNorm fused with the residual add + ReLU, 3x3/s2 max-pool.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as FD
from .. import tuning

_SPEC = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
         101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}


def _conv(x, conv, in_norm=False):
    return FD.conv2d(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0], in_norm=in_norm)


def _conv_bn(x, conv, bn, residual=None, relu=False, tap=False):
    """bn(conv(x)) [+ residual] [ReLU].  In training mode the convolution's epilogue gathers the BatchNorm's partial sums where its
    kernel can (FD.conv2d_stats), so the BatchNorm is ONE launch over the output instead of a statistics pass + an apply pass.
    ``tap``: also return ``x`` routed through the convolution's autograd node (see ``_conv_tap``)."""
    if bn.training and tuning.host.conv_stats and torch.is_grad_enabled():
        if tuning.host.fused_conv_bn and conv.bias is None:
            return FD.conv_bn(x, conv.weight, bn, stride=conv.stride[0], pad=conv.padding[0], residual=residual, relu=relu, tap=tap)
        res = FD.conv2d_stats(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0], tap=tap)
        y, stats = res[0], res[1]
        out = FD.batch_norm(y, bn, residual=residual, relu=relu, conv_stats=stats)
        return (out, res[2]) if tap else out
    if (not bn.training and residual is None and conv.bias is None and tuning.host.fold_frozen_bn and not torch.is_grad_enabled()
            and getattr(conv.weight, "_fd_frozen", False) and bn.weight is not None):
        y = FD.conv_bn_frozen(x, conv.weight, bn, stride=conv.stride[0], pad=conv.padding[0], relu=relu)   # frozen network: BN folded
        return (y, x) if tap else y
    if tap:
        y, x = _conv_tap(x, conv)
        return FD.batch_norm(y, bn, residual=residual, relu=relu), x
    return FD.batch_norm(_conv(x, conv), bn, residual=residual, relu=relu)


def _conv_tap(x, conv):
    """(conv(x), x): the block input is needed twice - by the first convolution and by the residual branch.  Taking the second
    use from the tap makes the residual gradient join the first convolution's data gradient inside that kernel
    (functional._Conv2dTap) instead of being summed by a separate element-wise launch."""
    return FD.conv2d_tap(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0])


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        out, x = _conv_bn(x, self.conv1, self.bn1, relu=True, tap=True)
        identity = x
        if self.downsample is not None:
            identity = _conv_bn(x, self.downsample[0], self.downsample[1])
        return _conv_bn(out, self.conv2, self.bn2, residual=identity, relu=True)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)     # v1.5: stride on the 3x3
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        out, x = _conv_bn(x, self.conv1, self.bn1, relu=True, tap=True)
        identity = x
        if self.downsample is not None:
            identity = _conv_bn(x, self.downsample[0], self.downsample[1])
        out = _conv_bn(out, self.conv2, self.bn2, relu=True)
        return _conv_bn(out, self.conv3, self.bn3, residual=identity, relu=True)


class _Stage(nn.Sequential):
    def forward(self, x):
        for blk in self:
            x = blk(x)
        return x


class ResNetTrunk(nn.Module):
    """Module tree of torchvision's ``ResNet`` (conv1, bn1, layer1..4, fc)."""

    def __init__(self, num_layers, in_channels):
        super().__init__()
        kind, counts = _SPEC[num_layers]
        block = BasicBlock if kind == "basic" else Bottleneck
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        inplanes = 64
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), counts)):
            blocks = []
            for bi in range(n):
                blocks.append(block(inplanes, planes, (1 if li == 0 else 2) if bi == 0 else 1))
                inplanes = planes * block.expansion
            setattr(self, "layer%d" % (li + 1), _Stage(*blocks))
        self.fc = nn.Linear(inplanes, 1000)       # never used (resnet_encoder.py:92-103); kept for checkpoint parity
        for m in self.modules():                  # torchvision / resnet_encoder.py:25-30 initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def stem_channels(num_input_images, cat4beam_to_color, cat2channel, beam_encoder, refine_encoder):
    """resnet_encoder.py:76-87: which conv1 the flags select."""
    if cat4beam_to_color:
        return 4
    if cat2channel:
        return 5
    if beam_encoder:
        return num_input_images * 2 if num_input_images > 1 else 2
    if refine_encoder:
        return 6
    return 3 * num_input_images


class ResnetEncoder(nn.Module):
    """Same constructor/forward contract as the reference (resnet_encoder.py:56-57,92-103)."""

    def __init__(self, num_layers, pretrained, num_input_images=1, cat4beam_to_color=False, cat2channel=False,
                 beam_encoder=False, refine_encoder=False):
        super().__init__()
        if num_layers not in _SPEC:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if pretrained:
            raise RuntimeError("ImageNet weights cannot be downloaded here (no network); run with "
                               "--weights_init scratch or load a checkpoint via load_state_dict")
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetTrunk(num_layers, stem_channels(num_input_images, cat4beam_to_color, cat2channel,
                                                              beam_encoder, refine_encoder))
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4
        self.stem_feature_needed = True       # False: features[0] is None in training mode (nobody reads it; saves its write + re-reads)

    def stem(self, input_image):
        """normalise, conv1, bn1, ReLU -> features[0]; max-pool -> the input of layer1 (resnet_encoder.py:94-98)"""
        e = self.encoder
        # (x - 0.45) / 0.225 (resnet_encoder.py:94) as its own pass: the 7x7 stem then gathers plain values.  An input that carries
        # ``_fd_normalized`` was normalised by its producer (Trainer._stack_pose_inputs: FD.stack_normalize assembles and normalises
        # the pose networks' input in one launch).
        xin = input_image if getattr(input_image, "_fd_normalized", False) else FD.input_normalize(input_image)
        x = FD.conv2d(xin, e.conv1.weight, None, stride=2, pad=3)
        if tuning.host.fused_stem_tail and e.bn1.training:
            # BatchNorm + ReLU + max-pool of the stem in one pass; features[0] is materialised only for the encoders whose skip
            # connection reads it (``stem_feature_needed``: the trainer clears it for the pose encoders - PoseDecoder reads features[-1])
            f0, x = FD.bn_relu_maxpool(x, e.bn1, want_feature=self.stem_feature_needed)
        else:
            f0 = FD.batch_norm(x, e.bn1, relu=True)
            x = FD.max_pool3x3s2(f0)
        return f0, x

    def forward_steps(self, input_image):
        """The forward pass as a generator that yields after the stem and after every residual block, so that a caller can
        advance several encoders in turn (``interleaved_forward``).  ``self.features`` is set when it is exhausted."""
        e = self.encoder
        f0, x = self.stem(input_image)
        yield
        feats = [f0]
        for li in range(1, 5):
            for blk in getattr(e, "layer%d" % li):
                x = blk(x)
                yield
            feats.append(x)
        self.features = feats

    def forward(self, input_image):
        for _ in self.forward_steps(input_image):
            pass
        return self.features


def interleaved_forward(jobs):
    """Advance several independent encoders block by block in round-robin order.  ``jobs``: list of (encoder, input, stream or
    None, BatchNorm groups).  The four encoders of a training step run on four HIP streams, but one Python thread issues them: issued
    one after the other, stream k only starts once the k-1 encoders before it have been issued in full, and - because autograd
    replays nodes in reverse creation order - the backward passes are issued one whole encoder at a time as well, the largest
    (first issued) last.  Issued in turns, all streams have work from the first block on, in both directions."""
    gens = [enc.forward_steps(x) for enc, x, _, _ in jobs]
    alive = list(range(len(jobs)))
    while alive:
        for i in list(alive):
            enc, _, st, groups = jobs[i]
            with FD.bn_groups(groups):
                try:
                    if st is None:
                        next(gens[i])
                    else:
                        with torch.cuda.stream(st):
                            next(gens[i])
                except StopIteration:
                    alive.remove(i)
    return [enc.features for enc, _, _, _ in jobs]
