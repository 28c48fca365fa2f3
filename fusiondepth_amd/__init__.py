"""fusiondepth_amd — MI355X-native (gfx950) implementation of FusionDepth's self-supervised training
hot path, behind the reference's ``layers`` / ``networks`` / ``options`` / ``trainer`` module API.

All arithmetic runs in hand-written HIP kernels (``csrc/*.hip`` -> ``libfdhip.so``, C ABI in
``include/fdhip.h``); this package is the host-side mirror of the reference interface.
There is no CPU fallback: ops raise if the library is not built or tensors are not on the GPU.
"""
__version__ = "0.1.0"
