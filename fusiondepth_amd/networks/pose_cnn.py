"""HIP-backed ``PoseCNN`` (reference networks/pose_cnn.py:7-44; non-default ``--pose_model_type posecnn``)."""
import torch.nn as nn

from .. import functional as FD


class PoseCNN(nn.Module):
    def __init__(self, num_input_frames):
        super().__init__()
        self.num_input_frames = num_input_frames
        spec = [(3 * num_input_frames, 16, 7, 3), (16, 32, 5, 2), (32, 64, 3, 1), (64, 128, 3, 1), (128, 256, 3, 1),
                (256, 256, 3, 1), (256, 256, 3, 1)]
        self.convs = {i: nn.Conv2d(ci, co, k, 2, p) for i, (ci, co, k, p) in enumerate(spec)}
        self.pose_conv = nn.Conv2d(256, 6 * (num_input_frames - 1), 1)
        self.num_convs = len(self.convs)
        self.relu = nn.ReLU(True)
        self.net = nn.ModuleList(list(self.convs.values()))

    def forward(self, out):
        for i in range(self.num_convs):
            c = self.convs[i]
            out = FD.conv2d(out, c.weight, c.bias, stride=2, pad=c.padding[0], act="relu")
        out = FD.conv2d(out, self.pose_conv.weight, self.pose_conv.bias, stride=1, pad=0)
        out = FD.spatial_mean(out, 0.01).view(-1, self.num_input_frames - 1, 1, 6)
        return out[..., :3], out[..., 3:]
