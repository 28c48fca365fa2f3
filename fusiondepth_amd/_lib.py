"""ctypes binding of ``libfdhip.so`` (C ABI declared in ``include/fdhip.h``).

There is no CPU fallback: every op in this package goes through this library and raises if it is not
built (``python -m fusiondepth_amd.build``) or if a tensor is not a contiguous float32 CUDA tensor.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FD_LIBFDHIP") or os.path.join(_HERE, "libfdhip.so")   # FD_LIBFDHIP: ablation builds (scripts/)
ABI_VERSION = 4
PHOTO_OUT_FLOATS = 96        # FD_PHOTO_OUT_FLOATS

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double
_KIND = {"p": _P, "i": _I, "l": _L, "f": _F, "d": _D}


class PhotoCfg(ctypes.Structure):
    """Mirror of ``fd_photo_cfg``."""
    _fields_ = [("min_depth", _D), ("max_depth", _D),
                ("B", _I), ("H", _I), ("W", _I), ("Hs", _I), ("Ws", _I), ("NF", _I),
                ("use_ssim", _I), ("avg_reprojection", _I),
                ("si_depth_scale", _F), ("si_beam_scale", _F), ("si_threshold", _F), ("si_var", _F), ("eps", _F),
                ("groups", _I), ("si_lo", _F), ("si_mode", _I)]


class PhotoMsCfg(ctypes.Structure):
    """Mirror of ``fd_photo_ms_cfg``."""
    _fields_ = [("base", PhotoCfg), ("n_scales", _I), ("Hs", _I * 4), ("Ws", _I * 4), ("beam_mask", ctypes.c_uint),
                ("rows_per_strip", _I)]


class RefineCfg(ctypes.Structure):
    """Mirror of ``fd_refine_cfg``."""
    _fields_ = [("B", _I), ("H", _I), ("W", _I), ("n_scales", _I), ("Hs", _I * 4), ("Ws", _I * 4),
                ("crop_y0", _I), ("crop_y1", _I), ("crop_x0", _I), ("crop_x1", _I), ("min_depth", _D), ("max_depth", _D),
                ("catxy", _I), ("pool_disp0", _I)]


class ConvDesc(ctypes.Structure):
    """Mirror of ``fd_conv_desc``."""
    _fields_ = [(n, _I) for n in ("N", "Cin", "H", "W", "Cout", "KH", "KW", "stride", "pad", "pad_mode", "act", "in_norm")]


class RelayoutJob(ctypes.Structure):
    """Mirror of ``fd_relayout_job``."""
    _fields_ = ([("w", ctypes.c_void_p), ("dst", ctypes.c_void_p)] +
                [(n, _I) for n in ("Co", "Ci", "KH", "KW", "TA", "TB", "kh0", "dkh", "kw0", "dkw", "mode", "reserved")] +
                [("n", ctypes.c_long), ("first_block", ctypes.c_long)])


# name -> (argument kinds, restype kind)  ('p' pointer, 'i' int, 'l' long, 'f' float, 'd' double)
SIGNATURES = {
    "fd_abi_version": ("", "i"),
    "fd_supported_arch": ("", "s"),
    "fd_last_error": ("", "s"),
    "fd_tuning_defaults": ("p", "v"),
    "fd_set_tuning": ("p", "i"),
    "fd_get_tuning": ("p", "v"),
    "fd_tuning_generation": ("", "l"),
    "fd_disp_to_depth_fwd": ("ppplddp", "i"),
    "fd_disp_to_depth_bwd": ("pppplddp", "i"),
    "fd_pose_matrix_fwd": ("pppiip", "i"),
    "fd_pose_matrix_bwd": ("pppppiip", "i"),
    "fd_pose_head_fwd": ("ppppiiiiip", "i"),
    "fd_pose_head_bwd": ("pppiiiiip", "i"),
    "fd_proj_matrix_fwd": ("ppplip", "i"),
    "fd_proj_matrix_bwd": ("pplpip", "i"),
    "fd_backproject_fwd": ("pppiiip", "i"),
    "fd_backproject_bwd": ("pppiiip", "i"),
    "fd_project3d_fwd": ("ppppiiifp", "i"),
    "fd_project3d_bwd_ws_floats": ("iii", "l"),
    "fd_project3d_bwd": ("pppppppiiifp", "i"),
    "fd_cat_xy_fwd": ("pppiiip", "i"),
    "fd_bilinear_up_fwd": ("ppiiiiip", "i"),
    "fd_bilinear_up_bwd": ("ppiiiiip", "i"),
    "fd_ssim_fwd": ("pppiiiip", "i"),
    "fd_ssim_bwd": ("pppppiiiip", "i"),
    "fd_reproj_loss_map": ("pppliiiip", "i"),
    "fd_photo_ws_floats": ("iii", "l"),
    "fd_photo_fwd": ("p" * 16, "i"),
    "fd_photo_bwd_ws_floats": ("iii", "l"),
    "fd_photo_bwd": ("pppppppp" "i" "pppppp", "i"),
    "fd_photo_fwd_ex": ("p" * 18, "i"),
    "fd_photo_bwd_ex": ("ppppppppp" "i" "pppppp", "i"),
    "fd_photo_ms_ws_floats": ("p", "l"),
    "fd_photo_ms_fwd": ("p" * 14, "i"),
    "fd_photo_ms_bwd": ("p" * 11, "i"),
    "fd_smooth_ws_floats": ("iii", "l"),
    "fd_smooth_fwd": ("ppppiiiip", "i"),
    "fd_smooth_bwd": ("pppppiiiip", "i"),
    "fd_scatter_2channel": ("ppiiiiiiiip", "i"),
    "fd_conv2d_fwd_wt_floats": ("p", "l"),
    "fd_conv2d_fwd_ws_floats": ("p", "l"),
    "fd_conv2d_fwd": ("pppppp" "i" "pp", "i"),
    "fd_conv2d_fwd_stat_slots": ("p", "l"),
    "fd_conv2d_fwd_stats": ("pppppp" "i" "ppp", "i"),
    "fd_conv2d_fwd_bn_ok": ("pi", "i"),
    "fd_conv2d_fwd_bn": ("pppppip" "pppppppp" "iffip", "i"),
    "fd_bn_train_fwd_parts": ("pppppppppp" "iiiiii" "ff" "ip", "i"),
    "fd_conv2d_bwd_data_wt_floats": ("p", "l"),
    "fd_conv2d_bwd_data_ws_floats": ("p", "l"),
    "fd_conv2d_bwd_data": ("ppppp" "i" "pp", "i"),
    "fd_conv2d_bwd_data_add": ("pppppp" "i" "pp", "i"),
    "fd_conv2d_bwd_data_inact": ("pppp" "i" "pp" "i" "pp", "i"),
    "fd_conv3x3_wino_wt_floats": ("p", "l"),
    "fd_conv3x3_wino_ws_floats": ("p", "l"),
    "fd_conv3x3_wino_fwd": ("pppppp" "i" "pp", "i"),
    "fd_velo_rasterize_ws_bytes": ("iii", "l"),
    "fd_velo_rasterize": ("pipiiiiipppp", "i"),
    "fd_resize_linear_cv": ("ppliiiip", "i"),
    "fd_masked_median_ws_bytes": ("iii", "l"),
    "fd_masked_median": ("ppfiiiiiiippp", "i"),
    "fd_refine_inputs_ws_bytes": ("p", "l"),
    "fd_refine_inputs": ("ppppppppp", "i"),
    "fd_conv2d_relayout_jobs": ("pippp", "i"),
    "fd_relayout_plan": ("pi", "l"),
    "fd_relayout_batch": ("pilp", "i"),
    "fd_conv2d_bwd_weight_ws_floats": ("p", "l"),
    "fd_conv2d_bwd_weight": ("pppppp" "i" "p", "i"),
    "fd_act_bwd": ("ppplip", "i"),
    "fd_bn_ws_floats": ("iiiii", "l"),
    "fd_bn_train_fwd": ("pppppppppp" "iiiii" "ff" "ip", "i"),
    "fd_bn_eval_fwd": ("ppppppp" "iiii" "f" "ip", "i"),
    "fd_bn_train_bwd": ("ppppppppppp" "iiiii" "iip", "i"),
    "fd_bn_train_bwd_remask": ("pppppppppp" "iiiii" "ip", "i"),
    "fd_bn_relu_maxpool_fwd": ("ppppppppppp" "iiiii" "ffp", "i"),
    "fd_bn_relu_maxpool_bwd": ("pppppppppppp" "iiiii" "ip", "i"),
    "fd_maxpool3x3s2_fwd": ("pppiiiip", "i"),
    "fd_maxpool3x3s2_bwd": ("pppiiiip", "i"),
    "fd_upcat_fwd": ("ppppp" "iiiiii" "p", "i"),
    "fd_upcat_bwd": ("pppp" "iiiiii" "p", "i"),
    "fd_upcat_bwd_act": ("pp" "i" "ppp" "iiiiii" "p", "i"),
    "fd_upsample2x_fwd": ("ppliip", "i"),
    "fd_upsample2x_bwd": ("ppliip", "i"),
    "fd_axpby": ("ppplffp", "i"),
    "fd_input_normalize": ("pplffp", "i"),
    "fd_stack_normalize": ("pppiiiiiipiffp", "i"),
    "fd_combine_losses_fwd": ("pppifpp", "i"),
    "fd_combine_losses_bwd": ("pifpp", "i"),
    "fd_spatial_mean_fwd": ("ppllfp", "i"),
    "fd_spatial_mean_bwd": ("ppllfp", "i"),
    "fd_depth_errors": ("pplppp", "i"),
    "fd_post_process_disparity": ("ppplii" "p", "i"),
    "fd_adam_step": ("ppppl" "fffffff" "p", "i"),
    "fd_adam_step_dev": ("ppppl" "p" "ffff" "p", "i"),
    "fd_replay_function_count": ("", "i"),
    "fd_replay_function_name": ("i", "s"),
    "fd_replay_function_signature": ("i", "s"),
    "fd_replay": ("pippip", "i"),
}

_lock = threading.Lock()
_lib = None


def load():
    """Load (once) and return the ctypes handle; raise if the library is missing or mismatched."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libfdhip.so is not built (%s missing). Run `python -m fusiondepth_amd.build` "
                               "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (args, res) in SIGNATURES.items():
            fn = getattr(lib, name)           # AttributeError if the symbol is not exported
            fn.argtypes = [_KIND[k] for k in args]
            fn.restype = ctypes.c_char_p if res == "s" else (None if res == "v" else _KIND[res])
        if lib.fd_abi_version() != ABI_VERSION:
            raise RuntimeError("libfdhip ABI %d != expected %d; rebuild" % (lib.fd_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def last_error():
    return load().fd_last_error().decode()


# tuning.host.host_delay_us = x: busy-wait x microseconds before every entry-point call - the experiment behind DESIGN.md's "the
# step is GPU-bound": up to 16 us per call (about +11 ms of host work per step) leaves ms_per_step unchanged (profiles/README.md,
# round 3).  Set by fusiondepth_amd/tuning.py.
HOST_DELAY_US = 0.0


# replay.py: while a region is being recorded, every entry-point call and every tensor handed to ``ptr`` is reported to the recorder
# (one list lookup per call otherwise); HOST_PERSISTENT = host addresses that may appear as literal pointer arguments of a recorded
# call (the shape-keyed convolution descriptors, kept alive by functional._CONV_PLANS and by the plans that reference them)
RECORDER = [None]
HOST_PERSISTENT = set()


def call(name, *args):
    """Invoke an ``int``-returning entry point and raise RuntimeError on a non-zero status."""
    lib = load()
    if RECORDER[0] is not None and name != "fd_replay":
        RECORDER[0].saw_call(name, args)
    if HOST_DELAY_US:
        import time
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e6 < HOST_DELAY_US:
            pass
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (status %d): %s" % (name, rc, lib.fd_last_error().decode()))


def query(name, *args):
    """Invoke a size-query entry point."""
    return getattr(load(), name)(*args)


def ptr(t):
    """Device pointer of a contiguous float32 (or uint8) CUDA tensor; ``None`` -> NULL."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("fusiondepth_amd ops run on the GPU only (got a %s tensor); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("tensor must be contiguous")
    if t.dtype not in (torch.float32, torch.uint8):
        raise RuntimeError("tensor must be float32 (or uint8 for masks), got %s" % t.dtype)
    if RECORDER[0] is not None:
        RECORDER[0].saw_tensor(t)
    return t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_GET_DEVICE = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """hipStream_t of the current PyTorch stream on the current device.  Called once per kernel launch (~1 400 times per
    training step): torch.cuda.current_stream() builds a Python Stream object (~10 us); the raw getters are plain C calls."""
    if _RAW_STREAM is not None and _GET_DEVICE is not None:
        return _RAW_STREAM(_GET_DEVICE())
    return torch.cuda.current_stream().cuda_stream


def f32(t):
    """Contiguous float32 view/copy (plumbing only)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
