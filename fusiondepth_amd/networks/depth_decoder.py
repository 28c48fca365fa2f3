"""HIP-backed ``DepthDecoder`` (reference networks/depth_decoder.py:6-96): same constructor flags, forward
signature, output dict and ``decoder.{i}.conv.conv.{weight,bias}`` / ``decoder.{10+s}.conv.*`` state-dict
layout.  Per stage: ConvBlock (reflect-pad 3x3 + ELU in the conv epilogue) -> one gather kernel doing the
nearest x2 upsample + skip concat + RGB/beam feature add -> ConvBlock; dispconv + sigmoid/tanh epilogue."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import functional as FD
from ..layers import Conv3x3, ConvBlock


class _ConvChain(nn.Sequential):
    """``deep`` refiner variant: Sequential(ConvBlock, ConvBlock) (depth_decoder.py:29-32,45-48)."""


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, cat2end=False, road=False,
                 catxy=False, deep=False):
        super().__init__()
        self.num_output_channels = num_output_channels
        self.use_skips = use_skips
        self.upsample_mode = "nearest"
        self.scales = scales
        self.num_ch_enc = num_ch_enc
        self.num_ch_dec = np.array([16, 32, 64, 128, 256])
        self.cat2end = cat2end

        def block(cin, cout):
            if not deep:
                return ConvBlock(cin, cout)
            return _ConvChain(ConvBlock(cin, cin), ConvBlock(cin, cout))

        self.convs = OrderedDict()
        for i in range(4, -1, -1):
            cin = self.num_ch_enc[-1] if i == 4 else self.num_ch_dec[i + 1]
            self.convs[("upconv", i, 0)] = block(cin, self.num_ch_dec[i])
            cin = self.num_ch_dec[i]
            if self.use_skips and i > 0:
                cin += self.num_ch_enc[i - 1]
            if road and i in self.scales and self.use_skips:
                cin += 6 if catxy else 3
            self.convs[("upconv", i, 1)] = block(cin, self.num_ch_dec[i])
        for s in self.scales:
            self.convs[("dispconv", s)] = Conv3x3(self.num_ch_dec[s], self.num_output_channels)
        if self.cat2end:
            self.convs[("dispconv", 0)] = Conv3x3(self.num_ch_dec[0] + 2, self.num_output_channels)
        self.decoder = nn.ModuleList(list(self.convs.values()))
        self.sigmoid = nn.Sigmoid()

    @staticmethod
    def _disp(conv3x3, x, act):
        c = conv3x3.conv
        return FD.conv2d(x, c.weight, c.bias, stride=1, pad=1, pad_mode="reflect" if conv3x3.use_refl else "zero", act=act)

    def forward(self, input_features, two_channel=None, beam_features=None, depth_maps=None, tanh=False):
        self.outputs = {}
        if beam_features is not None:
            x = FD.add(input_features[-1], beam_features[-1])
        else:
            x = input_features[-1]
        for i in range(4, -1, -1):
            x = self.convs[("upconv", i, 0)](x)
            skip = skip_add = extra = None
            if self.use_skips and i > 0:
                skip = input_features[i - 1]
                if beam_features is not None:
                    skip_add = beam_features[i - 1]
            if depth_maps is not None and i in self.scales and self.use_skips:
                extra = depth_maps[("disp", i)]
            x = FD.upsample_concat(x, skip, skip_add, extra)
            x = self.convs[("upconv", i, 1)](x)
            if i in self.scales:
                if i == 0 and self.cat2end:
                    self.outputs[("disp", i)] = self._disp(self.convs[("dispconv", i)], torch.cat((x, two_channel), 1),
                                                           "sigmoid")
                else:
                    self.outputs[("disp", i)] = self._disp(self.convs[("dispconv", i)], x, "tanh" if tanh else "sigmoid")
        return self.outputs
