"""HIP-backed ``PoseDecoder`` (reference networks/pose_decoder.py:8-51): 1x1 squeeze + ReLU, two 3x3 + ReLU,
1x1 -> 6*nf, spatial mean, x0.01.  State-dict keys ``net.{0..3}.{weight,bias}``."""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import functional as FD


def pose_layer_table(enc_width, num_input_features, num_frames, stride):
    """(key, Cin, Cout, kernel, stride, pad) in state-dict order ``net.0 .. net.3`` (pose_decoder.py:19-27)."""
    return [("squeeze", int(enc_width), 256, 1, 1, 0),
            (("pose", 0), num_input_features * 256, 256, 3, stride, 1),
            (("pose", 1), 256, 256, 3, stride, 1),
            (("pose", 2), 256, 6 * num_frames, 1, 1, 0)]


class PoseDecoder(nn.Module):
    def __init__(self, num_ch_enc, num_input_features, num_frames_to_predict_for=None, stride=1):
        super().__init__()
        self.num_ch_enc, self.num_input_features = num_ch_enc, num_input_features
        self.num_frames_to_predict_for = (num_input_features - 1 if num_frames_to_predict_for is None
                                          else num_frames_to_predict_for)
        self.convs = OrderedDict((key, nn.Conv2d(cin, cout, k, st, pad)) for key, cin, cout, k, st, pad in
                                 pose_layer_table(num_ch_enc[-1], num_input_features, self.num_frames_to_predict_for, stride))
        self.relu = nn.ReLU()
        self.net = nn.ModuleList(self.convs.values())

    @staticmethod
    def _conv(x, conv, act):
        return FD.conv2d(x, conv.weight, conv.bias, stride=conv.stride[0], pad=conv.padding[0], act=act)

    def forward(self, input_features, beam_inputs=None, raw=False):
        if beam_inputs is not None:                          # RGB + LiDAR bottleneck features, one input (pose_decoder.py:30-31)
            deepest = [FD.add(input_features[0][-1], beam_inputs[0][-1])]
        else:
            deepest = [feats[-1] for feats in input_features]
        squeezed = [self._conv(f, self.convs["squeeze"], "relu") for f in deepest]
        out = squeezed[0] if len(squeezed) == 1 else torch.cat(squeezed, 1)
        for i, act in enumerate(("relu", "relu", "none")):
            out = self._conv(out, self.convs[("pose", i)], act)
        out = FD.spatial_mean(out, 0.01)                     # 0.01 * out.mean(3).mean(2)
        if raw:                                              # [B, 6 * frames]: the trainer's fused pose head (FD.pose_head) slices it itself
            return out
        out = out.view(-1, self.num_frames_to_predict_for, 1, 6)
        return out[..., :3], out[..., 3:]
