"""Oracle restatement of reference ``networks/*`` (fp32 CPU ``torch.nn`` modules).

State-dict key names and shapes equal the reference's so the
same weights load into reference modules, oracle modules and the HIP-backed
product modules.

``ResNetTrunk`` restates torchvision-0.9's ResNet v1.5 (BasicBlock / Bottleneck
with the stride on the 3x3, downsample = conv1x1(stride)+BN, BN eps 1e-5,
momentum 0.1).  torchvision is a third-party dependency of the reference
(networks/resnet_encoder.py:7,62-74) that is neither vendored nor installed
here; the restatement is pinned against an independent public implementation of the
same architecture that is installed - ``transformers.ResNetModel`` - with shared weights
(tests/test_oracle_golden.py::test_resnet_trunk_matches_an_independent_resnet: feature
maps and input gradients, ResNet-18 / -50, train and eval mode) plus the structural
checks (torchvision's state-dict keys, shapes and parameter counts).
"""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import layers as L

_BLOCKS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottle", [3, 4, 6, 3]),
           101: ("bottle", [3, 4, 23, 3]), 152: ("bottle", [3, 8, 36, 3])}


class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or cin != planes:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + idn)


class _Bottle(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or cin != planes * 4:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes * 4, 1, stride, bias=False),
                                            nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idn = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + idn)


class ResNetTrunk(nn.Module):
    """torchvision ResNet body (keys conv1, bn1, layer1-4, fc)."""

    def __init__(self, num_layers, in_channels=3, multi_image_init=False):
        super().__init__()
        kind, counts = _BLOCKS[num_layers]
        block = _Basic if kind == "basic" else _Bottle
        self.conv1 = nn.Conv2d(in_channels, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for li, (planes, n) in enumerate(zip([64, 128, 256, 512], counts)):
            blocks = []
            for bi in range(n):
                blocks.append(block(cin, planes, (1 if li == 0 else 2) if bi == 0 else 1))
                cin = planes * block.expansion
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, 1000)      # unused by the encoder; kept for checkpoint-key parity
        for m in self.modules():            # torchvision / resnet_encoder.py:25-30 init
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def conv1_in_channels(num_input_images=1, cat4beam_to_color=False, cat2channel=False, beam_encoder=False,
                      refine_encoder=False):
    """resnet_encoder.py:76-87 — channel count of the swapped stem."""
    if cat4beam_to_color:
        return 4
    if cat2channel:
        return 5
    if beam_encoder:
        return 2 * num_input_images if num_input_images > 1 else 2
    if refine_encoder:
        return 6
    return 3 * num_input_images


class ResnetEncoder(nn.Module):
    """networks/resnet_encoder.py:53-103."""

    def __init__(self, num_layers, pretrained, num_input_images=1, cat4beam_to_color=False, cat2channel=False,
                 beam_encoder=False, refine_encoder=False):
        super().__init__()
        if pretrained:
            raise RuntimeError("ImageNet weights are not available offline; use weights_init=scratch")
        if num_layers not in _BLOCKS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        cin = conv1_in_channels(num_input_images, cat4beam_to_color, cat2channel, beam_encoder, refine_encoder)
        self.encoder = ResNetTrunk(num_layers, cin)
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def forward(self, input_image):
        e = self.encoder
        x = (input_image - 0.45) / 0.225
        f0 = e.relu(e.bn1(e.conv1(x)))
        f1 = e.layer1(e.maxpool(f0))
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features


class _Conv3x3(nn.Module):
    def __init__(self, cin, cout, use_refl=True):
        super().__init__()
        self.use_refl = use_refl
        self.conv = nn.Conv2d(int(cin), int(cout), 3)

    def forward(self, x):
        return L.conv3x3(x, self.conv.weight, self.conv.bias, self.use_refl)


class _ConvBlock(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = _Conv3x3(cin, cout)

    def forward(self, x):
        return F.elu(self.conv(x))


class DepthDecoder(nn.Module):
    """networks/depth_decoder.py:6-96."""

    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, cat2end=False,
                 road=False, catxy=False, deep=False):
        super().__init__()
        self.scales = scales
        self.use_skips = use_skips
        self.cat2end = cat2end
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.convs = OrderedDict()

        def block(ci, co):
            if not deep:
                return _ConvBlock(ci, co)
            return nn.Sequential(_ConvBlock(ci, ci), _ConvBlock(ci, co))

        for i in range(4, -1, -1):
            ci = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = block(ci, self.num_ch_dec[i])
            ci = self.num_ch_dec[i]
            if use_skips and i > 0:
                ci += self.num_ch_enc[i - 1]
            if road and i in self.scales and use_skips:
                ci += 6 if catxy else 3
            self.convs[("upconv", i, 1)] = block(ci, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = _Conv3x3(self.num_ch_dec[s], num_output_channels)
        if cat2end:
            self.convs[("dispconv", 0)] = _Conv3x3(self.num_ch_dec[0] + 2, num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features, two_channel=None, beam_features=None, depth_maps=None, tanh=False):
        def feat(i):
            if beam_features is not None:
                return input_features[i] + beam_features[i]
            return input_features[i]

        self.outputs = {}
        x = feat(4)
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            parts = [L.upsample(x)]
            if self.use_skips and i > 0:
                parts.append(feat(i - 1))
            if depth_maps is not None and i in self.scales and self.use_skips:
                parts.append(depth_maps[("disp", i)])
            x = self.convs[("upconv", i, 1)](torch.cat(parts, 1))
            if i in self.scales:
                if i == 0 and self.cat2end:
                    self.outputs[("disp", i)] = torch.sigmoid(self.convs[("dispconv", i)](torch.cat((x, two_channel), 1)))
                elif tanh:
                    self.outputs[("disp", i)] = torch.tanh(self.convs[("dispconv", i)](x))
                else:
                    self.outputs[("disp", i)] = torch.sigmoid(self.convs[("dispconv", i)](x))
        return self.outputs


class PoseDecoder(nn.Module):
    """networks/pose_decoder.py:8-51."""

    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs["squeeze"] = nn.Conv2d(self.num_ch_enc[-1], 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, input_features, beam_inputs=None):
        if beam_inputs is not None:
            last = [input_features[0][-1] + beam_inputs[0][-1]]
        else:
            last = [f[-1] for f in input_features]
        out = torch.cat([F.relu(self.convs["squeeze"](f)) for f in last], 1)
        for i in range(3):
            out = self.convs[("pose", i)](out)
            if i != 2:
                out = F.relu(out)
        out = out.mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]


class PoseCNN(nn.Module):
    """networks/pose_cnn.py:7-44."""

    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        spec = [(3 * num_input_frames, 16, 7, 3), (16, 32, 5, 2), (32, 64, 3, 1), (64, 128, 3, 1),
                (128, 256, 3, 1), (256, 256, 3, 1), (256, 256, 3, 1)]
        self.convs = {i: nn.Conv2d(ci, co, k, 2, p) for i, (ci, co, k, p) in enumerate(spec)}
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)
        self.num_convs = len(self.convs)
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, out):
        for i in range(self.num_convs):
            out = F.relu(self.convs[i](out))
        out = self.pose_conv(out).mean(3).mean(2)
        out = 0.01 * out.view(-1, self.num_input_frames - 1, 1, 6)
        return out[..., :3], out[..., 3:]
