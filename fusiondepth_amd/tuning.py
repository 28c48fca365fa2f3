"""Tuning switches, in ONE place.

Two kinds, both with the measured-best defaults in force when nothing is set:

* ``lib``: the library's kernel-selection thresholds (``fd_tuning`` in include/fdhip.h).  libfdhip never reads the process
  environment; ``set_lib(**fields)`` / ``with override(**fields):`` call ``fd_set_tuning``.  Workspace / weight-layout sizes depend
  on them, so ``generation()`` is part of functional.py's plan key.
* ``host``: how the Python side issues the step (streams, stacking, which loss kernel).  Plain attributes of ``host``.

For the A/B scripts under scripts/ (which set ``FD_*`` variables before they start Python) the environment is mapped onto both ONCE,
at import of this module - the only place in the package that looks at ``FD_*`` tuning variables.  Nothing reads the
environment afterwards: changing ``os.environ`` inside a running process has no effect (use ``set_lib`` / ``host``)."""
import contextlib
import ctypes
import os

from . import _lib

_I = ctypes.c_int


class Tuning(ctypes.Structure):
    """Mirror of ``fd_tuning`` (include/fdhip.h)."""
    _fields_ = [(n, _I) for n in (
        "size", "wino_fwd", "wino_wgrad", "wino_fwd_2d_min", "wino_fwd_2dp_min_wgs", "wino_fwd_2dp_dma", "wino_fwd_2dp_deep", "wino_wgrad_2d", "wino_target", "wino_wgrad_target", "conv_target",
        "wgrad_target", "conv_c1", "conv_n16_min_pixels", "reflect_ring", "reflect_wino", "reflect_wino_min_pixels",
        "reflect_wino_padded_max", "force_cfg", "force_splits", "stem7", "log", "wino_fwd_2d_m128", "wino_min_cout", "wino_wgrad_min_cout", "wino_wgrad_xcd_few", "wino_fwd_halfm", "wino_wgrad_halfm",
        "grp_tile64_below", "limb_1x1", "limb_depth", "limb_target", "limb_split_max_out", "limb_wgrad_target", "limb_conv", "wino_wgrad_limb", "wino_fwd_limb")]


LIB_FIELDS = tuple(n for n, _ in Tuning._fields_ if n != "size")


def lib_defaults():
    t = Tuning()
    _lib.load().fd_tuning_defaults(ctypes.byref(t))
    return t


def get_lib():
    t = Tuning()
    _lib.load().fd_get_tuning(ctypes.byref(t))
    return {n: getattr(t, n) for n in LIB_FIELDS}


def generation():
    """Number of fd_set_tuning calls so far (cached size queries are keyed on it).  Read from the library, so that a direct
    ``fd_set_tuning`` call through ctypes invalidates the cached plan sizes as well."""
    fn = _GEN_FN[0]
    if fn is None:
        fn = _GEN_FN[0] = _lib.load().fd_tuning_generation
    return int(fn())


_GEN_FN = [None]


def set_lib(**fields):
    """Change library thresholds (unknown names raise).  Returns the previous values of the fields changed."""
    t = Tuning()
    lib = _lib.load()
    lib.fd_get_tuning(ctypes.byref(t))
    prev = {}
    for k, v in fields.items():
        if k not in LIB_FIELDS:
            raise KeyError("fd_tuning has no field %r (fields: %s)" % (k, ", ".join(LIB_FIELDS)))
        prev[k] = getattr(t, k)
        setattr(t, k, int(v))
    t.size = ctypes.sizeof(Tuning)
    rc = lib.fd_set_tuning(ctypes.byref(t))
    if rc != 0:
        raise RuntimeError("fd_set_tuning failed: %s" % lib.fd_last_error().decode())
    return prev


@contextlib.contextmanager
def override(**fields):
    """``with tuning.override(wino_fwd=0): ...`` - library thresholds for the block (tests, sweeps), restored on exit."""
    prev = set_lib(**fields)
    try:
        yield
    finally:
        set_lib(**prev)


class _Host:
    """Host-side issue switches.  Defaults = the measured best."""
    late_relayout = True        # post-Adam re-layout of the large layouts on a side stream (-0.9 % images/s when off)
    pose_stream = True          # pose decoder on the pose encoder's stream (-3.2 % when off)
    smooth_stream = True        # smoothness terms beside the photometric kernel (-0.5 % when off)
    photo_ms = True             # the all-scales loss kernel for the default configuration (off: per-scale kernels)
    side_wgrad = ("depth", "refine2d_decoder")     # networks whose weight gradients run on a side stream (the steps' serial chains)
    n_streams = 8               # HIP streams of the training step (4 are used; 1 = everything on one stream)
    interleave = False          # the four encoders issued block by block in turns
    conv_stats = True           # BatchNorm statistics from the convolution epilogue where the kernel has one
    fused_conv_bn = True        # conv + BatchNorm of a ResNet block as ONE autograd node (host time only: same launches)
    refiner_streams = True      # the Refiner's frozen encoders on per-module streams
    fold_frozen_bn = True       # eval-mode BatchNorm of a frozen ResNet block folded into its convolution (no residual input)
    refiner_prefetch = True     # Refiner.train_step(inputs, next_inputs): the next batch's frozen forward passes overlap this step
    decoder_fused_act = True    # ELU' of the decoder's single-consumer blocks applied where the gradient is produced (-0.15 ms when on)
    dp_overlap = True           # per-network gradient buckets all-reduced from inside the backward pass
    fused_finish_bn = True      # deep-layer conv + BatchNorm: the F(2x2, 3x3) slab reduction inside the small-plane BatchNorm kernel
    bn_remask = True            # BatchNorm + ReLU without a residual: the backward recomputes the ReLU mask from x instead of reading y
    fused_stem_tail = True      # BatchNorm + ReLU + max-pool of the ResNet stems as one pass each way (csrc/norm.hip: k_bn_relu_pool_*)
    fused_pose_head = True      # stacked pose network: slicing + concatenation + pose matrices of all frame pairs as one launch each way (FD.pose_head)
    replay_frozen = True        # frozen no-grad sub-networks (the Refiner's stage-1 encoders / depth decoder) recorded once and replayed by ONE C call each (replay.py)
    replay_train = True         # the four ResNet encoders' TRAINING forward + backward as recorded call sequences behind one autograd node each (replay.TrainReplayable)
    early_loss_inputs = False   # identity reprojection losses + tie-break noise issued beside the encoders instead of between decoder and loss kernel
                                # (bit-identical; measured 665.1 vs 665.2 images/s same box, profiles/round6_early_loss_inputs_ab.log: off)
    pad_odd_channels = True     # refine decoder: blocks with 262 / 134 / 102 / 22 input channels run zero-padded to a multiple of 16

    @property
    def host_delay_us(self):    # busy-wait before every entry-point call (the host-slack experiment, profiles/round3_experiments.md)
        return _lib.HOST_DELAY_US

    @host_delay_us.setter
    def host_delay_us(self, v):  # lives in _lib (read by _lib.call): assigning the attribute at run time takes effect at once
        _lib.HOST_DELAY_US = float(v)


host = _Host()

# FD_* variable -> (kind, name, converter).  Read once, below.
_ENV_LIB = {
    "FD_WINO_FWD": ("wino_fwd", int), "FD_WINO_WGRAD": ("wino_wgrad", int), "FD_WINO_FWD_2D_MIN": ("wino_fwd_2d_min", int),
    "FD_WINO_WGRAD_2D": ("wino_wgrad_2d", int), "FD_WINO_FWD_2DP_MIN": ("wino_fwd_2dp_min_wgs", int), "FD_WINO_FWD_2DP_DMA": ("wino_fwd_2dp_dma", int), "FD_WINO_FWD_2DP_DEEP": ("wino_fwd_2dp_deep", int), "FD_WINO_TARGET": ("wino_target", int),
    "FD_WINO_WGRAD_TARGET": ("wino_wgrad_target", int), "FD_CONV_TARGET": ("conv_target", int), "FD_WGRAD_TARGET": ("wgrad_target", int),
    "FD_CONV_C1": ("conv_c1", int), "FD_CONV_N16_MIN": ("conv_n16_min_pixels", int), "FD_REFLECT_RING": ("reflect_ring", int),
    "FD_REFLECT_WINO": ("reflect_wino", int), "FD_REFLECT_WINO_MIN": ("reflect_wino_min_pixels", int),
    "FD_REFLECT_WINO_PADDED_MAX": ("reflect_wino_padded_max", int), "FD_STEM7": ("stem7", int), "FD_CONV_LOG": ("log", int), "FD_WINO_FWD_2D_M128": ("wino_fwd_2d_m128", int),
    "FD_WINO_MIN_COUT": ("wino_min_cout", int), "FD_WINO_WGRAD_MIN_COUT": ("wino_wgrad_min_cout", int), "FD_GRP_TILE64_BELOW": ("grp_tile64_below", int), "FD_LIMB_1X1": ("limb_1x1", int), "FD_LIMB_DEPTH": ("limb_depth", int), "FD_LIMB_TARGET": ("limb_target", int), "FD_LIMB_SPLIT_MAX_OUT": ("limb_split_max_out", int), "FD_LIMB_WGRAD_TARGET": ("limb_wgrad_target", int), "FD_LIMB_CONV": ("limb_conv", int), "FD_WINO_WGRAD_LIMB": ("wino_wgrad_limb", int), "FD_WINO_FWD_LIMB": ("wino_fwd_limb", int), "FD_WINO_WGRAD_XCD_FEW": ("wino_wgrad_xcd_few", int), "FD_WINO_WGRAD_HALFM": ("wino_wgrad_halfm", int), "FD_WINO_FWD_HALFM": ("wino_fwd_halfm", int),
}
_ENV_HOST = {
    "FD_LATE_RELAYOUT": ("late_relayout", lambda v: v != "0"), "FD_POSE_STREAM": ("pose_stream", lambda v: v != "0"),
    "FD_SMOOTH_STREAM": ("smooth_stream", lambda v: v != "0"), "FD_PHOTO_MS": ("photo_ms", lambda v: v != "0"),
    "FD_SIDE_WGRAD": ("side_wgrad", lambda v: tuple(k for k in v.split(",") if k and k != "none")),
    "FD_NSTREAMS": ("n_streams", int), "FD_INTERLEAVE": ("interleave", lambda v: v != "0"),
    "FD_CONV_STATS": ("conv_stats", lambda v: v != "0"), "FD_FUSED_CONV_BN": ("fused_conv_bn", lambda v: v != "0"), "FD_REFINER_STREAMS": ("refiner_streams", lambda v: v != "0"),
    "FD_REFINER_PREFETCH": ("refiner_prefetch", lambda v: v != "0"), "FD_FOLD_FROZEN_BN": ("fold_frozen_bn", lambda v: v != "0"),
    "FD_DP_OVERLAP": ("dp_overlap", lambda v: v != "0"), "FD_REPLAY_FROZEN": ("replay_frozen", lambda v: v != "0"), "FD_REPLAY_TRAIN": ("replay_train", lambda v: v != "0"), "FD_PAD_ODD_CHANNELS": ("pad_odd_channels", lambda v: v != "0"), "FD_EARLY_LOSS_INPUTS": ("early_loss_inputs", lambda v: v != "0"), "FD_FUSED_POSE_HEAD": ("fused_pose_head", lambda v: v != "0"), "FD_FUSED_STEM_TAIL": ("fused_stem_tail", lambda v: v != "0"), "FD_BN_REMASK": ("bn_remask", lambda v: v != "0"), "FD_FUSED_FINISH_BN": ("fused_finish_bn", lambda v: v != "0"), "FD_DECODER_FUSED_ACT": ("decoder_fused_act", lambda v: v != "0"), "FD_HOST_DELAY_US": ("host_delay_us", float),
}


def _from_environment():
    env = os.environ
    lib_fields = {}
    for var, (name, conv) in _ENV_LIB.items():
        if var in env:
            lib_fields[name] = conv(env[var])
    if env.get("FD_WINO") == "0":                     # everything on the direct implicit-GEMM kernels
        lib_fields.update(wino_fwd=0, wino_wgrad=0)
    if env.get("FD_WINO_FWD_2D") == "0":
        lib_fields["wino_fwd_2d_min"] = 0
    if env.get("FD_CONV_N16") == "0":
        lib_fields["conv_n16_min_pixels"] = -1
    if env.get("FD_REFLECT_WINO_PADDED") == "0":
        lib_fields["reflect_wino_padded_max"] = 0
    try:
        if "FD_CONV_FORCE" in env:                        # "cfg,splits"
            c, sp = env["FD_CONV_FORCE"].split(",")
            lib_fields.update(force_cfg=int(c), force_splits=int(sp))
        if lib_fields:
            set_lib(**lib_fields)
    except (ValueError, RuntimeError, KeyError) as e:
        raise RuntimeError("fusiondepth_amd.tuning: an FD_* variable of the A/B scripts has a value fd_set_tuning rejects (%s); "
                           "unset it or fix it - FD_* variables are only read here, once, at import" % e) from e
    for var, (name, conv) in _ENV_HOST.items():
        if var in env:
            setattr(host, name, conv(env[var]))


_from_environment()
