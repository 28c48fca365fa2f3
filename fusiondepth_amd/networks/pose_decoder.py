"""HIP-backed ``PoseDecoder`` (reference networks/pose_decoder.py:8-51): 1x1 squeeze + ReLU, two 3x3 + ReLU,
1x1 -> 6*nf, spatial mean, x0.01.  State-dict keys ``net.{0..3}.{weight,bias}``."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as FD


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc = num_ch_enc
        self.num_input_features = num_input_features
        if num_frames_to_predict_for is None:
            num_frames_to_predict_for = num_input_features - 1
        self.num_frames_to_predict_for = num_frames_to_predict_for
        self.convs = OrderedDict()
        self.convs[("squeeze")] = nn.Conv2d(self.num_ch_enc[-1], 256, 1)
        self.convs[("pose", 0)] = nn.Conv2d(num_input_features * 256, 256, 3, stride, 1)
        self.convs[("pose", 1)] = nn.Conv2d(256, 256, 3, stride, 1)
        self.convs[("pose", 2)] = nn.Conv2d(256, 6 * num_frames_to_predict_for, 1)
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(list(self.convs.values()))

    @staticmethod
    def _conv(x, conv, act):
        return FD.conv2d(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0], act=act)

    def forward(self, input_features, beam_inputs=None):
        if beam_inputs is not None:
            last_features = [FD.add(input_features[0][-1], beam_inputs[0][-1])]
        else:
            last_features = [f[-1] for f in input_features]
        cat_features = [self._conv(f, self.convs["squeeze"], "relu") for f in last_features]
        out = cat_features[0] if len(cat_features) == 1 else torch.cat(cat_features, 1)
        for i in range(3):
            out = self._conv(out, self.convs[("pose", i)], "relu" if i != 2 else "none")
        out = FD.spatial_mean(out, 0.01)                     # 0.01 * out.mean(3).mean(2)
        out = out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]
