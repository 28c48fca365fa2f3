"""HIP-backed ``DepthDecoder`` (reference networks/depth_decoder.py:6-96): same constructor flags, forward
signature, output dict and ``decoder.{i}.conv.conv.{weight,bias}`` / ``decoder.{10+s}.conv.*`` state-dict
layout.  Per stage: ConvBlock (reflect-pad 3x3 + ELU in the conv epilogue) -> one gather kernel doing the
nearest x2 upsample + skip concat + RGB/beam feature add -> ConvBlock; dispconv + sigmoid/tanh epilogue."""
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from .. import functional as FD
from .. import tuning
from ..layers import Conv3x3, ConvBlock


class _ConvChain(nn.Sequential):
    """``deep`` refiner variant: Sequential(ConvBlock, ConvBlock) (depth_decoder.py:29-32,45-48)."""


DEC_WIDTHS = (16, 32, 64, 128, 256)      # output channels of decoder level 0..4 (depth_decoder.py:19)


def decoder_layer_table(enc_widths, scales, out_channels, use_skips, cat2end, road, catxy):
    """The decoder as a table of (key, Cin, Cout) rows in state-dict order - ``decoder.<row index>`` is the checkpoint key
    the reference produces (depth_decoder.py:23-59): per level 4..0 the pair (upconv, i, 0), (upconv, i, 1), then one
    (dispconv, s) per scale; ``cat2end`` widens the full-resolution head by the two LiDAR channels."""
    rows = []
    for level in (4, 3, 2, 1, 0):
        below = enc_widths[-1] if level == 4 else DEC_WIDTHS[level + 1]
        rows.append((("upconv", level, 0), int(below), DEC_WIDTHS[level]))
        merged = DEC_WIDTHS[level]
        if use_skips and level > 0:
            merged += int(enc_widths[level - 1])
            if road and level in scales:
                merged += 6 if catxy else 3
        elif use_skips and road and level in scales:
            merged += 6 if catxy else 3
        rows.append((("upconv", level, 1), merged, DEC_WIDTHS[level]))
    heads = {s: DEC_WIDTHS[s] for s in scales}
    if cat2end:
        heads[0] = DEC_WIDTHS[0] + 2
    rows += [(("dispconv", s), cin, out_channels) for s, cin in heads.items()]
    return rows


class DepthDecoder(nn.Module):
    def __init__(self, num_ch_enc, scales=range(4), num_output_channels=1, use_skips=True, cat2end=False, road=False,
                 catxy=False, deep=False):
        super().__init__()
        self.num_ch_enc, self.num_ch_dec = num_ch_enc, np.array(DEC_WIDTHS)
        self.scales, self.num_output_channels = scales, num_output_channels
        self.use_skips, self.cat2end, self.upsample_mode = use_skips, cat2end, "nearest"
        self.convs = OrderedDict()
        for key, cin, cout in decoder_layer_table(num_ch_enc, scales, num_output_channels, use_skips, cat2end, road, catxy):
            if key[0] == "dispconv":
                self.convs[key] = Conv3x3(cin, cout)
            elif deep:
                self.convs[key] = _ConvChain(ConvBlock(cin, cin), ConvBlock(cin, cout))
            else:
                self.convs[key] = ConvBlock(cin, cout)
        self.decoder = nn.ModuleList(self.convs.values())
        self.sigmoid = nn.Sigmoid()
        # (upconv, i, 1) blocks whose input width is not a multiple of 16 (road variants: 262 / 134 / 102 / 22 channels): run
        # channel-padded (ConvBlock.forward(channels=...)): key -> padded input width
        self._cin_pad = {}
        if road:
            for key, cin, cout in decoder_layer_table(num_ch_enc, scales, num_output_channels, use_skips, cat2end, road, catxy):
                if key[0] == "upconv" and key[2] == 1 and cin % 16 != 0:
                    self._cin_pad[key] = (cin + 15) // 16 * 16

    @staticmethod
    def _block_cin(blk):
        return (blk[0] if isinstance(blk, _ConvChain) else blk).conv.conv.weight.shape[1]

    @staticmethod
    def _disp(conv3x3, x, act, in_act="none"):
        c = conv3x3.conv
        return FD.conv2d(x, c.weight, c.bias, stride=1, pad=1, pad_mode="reflect" if conv3x3.use_refl else "zero", act=act, in_act=in_act)

    def forward(self, input_features, two_channel=None, beam_features=None, depth_maps=None, tanh=False):
        self.outputs = {}
        if beam_features is not None:
            x = FD.add(input_features[-1], beam_features[-1])
        else:
            x = input_features[-1]
        # ELU' where the gradient is produced (FD.conv2d's grad_preact / in_act contract): upconv(i, 0) feeds the concatenation only,
        # and the full-resolution upconv(0, 1) feeds dispconv(0) only - their element-wise ELU-backward passes (0.15 ms of the step's
        # serial decoder -> loss -> decoder section) ride in fd_upcat_bwd_act / in the disparity head's data-gradient stencil
        fuse = torch.is_grad_enabled() and tuning.host.decoder_fused_act
        for i in range(4, -1, -1):
            blk0 = self.convs[("upconv", i, 0)]
            pre0 = fuse and isinstance(blk0, ConvBlock)
            x = blk0(x, grad_preact=True) if pre0 else blk0(x)
            skip = skip_add = extra = None
            if self.use_skips and i > 0:
                skip = input_features[i - 1]
                if beam_features is not None:
                    skip_add = beam_features[i - 1]
            if depth_maps is not None and i in self.scales and self.use_skips:
                extra = depth_maps[("disp", i)]
            blk1 = self.convs[("upconv", i, 1)]
            cin_p = self._cin_pad.get(("upconv", i, 1)) if (extra is not None and tuning.host.pad_odd_channels) else None
            if cin_p is not None:
                # zero channels behind the depth maps bring the concatenation to a multiple of 16; the block(s) run channel-padded
                extra = torch.nn.functional.pad(extra, (0, 0, 0, 0, 0, cin_p - self._block_cin(blk1)))
            x = FD.upsample_concat(x, skip, skip_add, extra, a_act="elu" if pre0 else "none")
            pre1 = fuse and i == 0 and i in self.scales and not self.cat2end and isinstance(blk1, ConvBlock)
            if cin_p is not None and isinstance(blk1, _ConvChain):
                x = blk1[0](x, channels=(cin_p, cin_p))
                x = blk1[1](x, channels=(cin_p, blk1[1].conv.conv.weight.shape[0]))
            elif cin_p is not None:
                x = blk1(x, grad_preact=pre1, channels=(cin_p, blk1.conv.conv.weight.shape[0]))
            else:
                x = blk1(x, grad_preact=True) if pre1 else blk1(x)
            if pre1:
                self.outputs[("disp", i)] = self._disp(self.convs[("dispconv", i)], x, "tanh" if tanh else "sigmoid", in_act="elu")
            elif i in self.scales:
                if i == 0 and self.cat2end:
                    self.outputs[("disp", i)] = self._disp(self.convs[("dispconv", i)], torch.cat((x, two_channel), 1),
                                                           "sigmoid")
                else:
                    self.outputs[("disp", i)] = self._disp(self.convs[("dispconv", i)], x, "tanh" if tanh else "sigmoid")
        return self.outputs
