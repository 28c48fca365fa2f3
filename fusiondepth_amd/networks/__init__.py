"""Drop-in for the reference ``networks`` package (networks/__init__.py:1-4)."""
from .resnet_encoder import ResnetEncoder, interleaved_forward
from .depth_decoder import DepthDecoder
from .pose_decoder import PoseDecoder
from .pose_cnn import PoseCNN

__all__ = ["ResnetEncoder", "DepthDecoder", "PoseDecoder", "PoseCNN"]
