"""``torch.autograd.Function`` wrappers over the libfdhip C ABI (loss path and layers.py ops).

Host logic only: shape checks, workspace allocation from PyTorch's caching allocator, and the
forward/backward pairing.  All arithmetic happens in the HIP kernels; nothing here falls back to
torch ops on the data path.
"""
import ctypes
import weakref

import torch

from . import _lib
from . import tuning
from ._lib import call, f32, ptr, query, stream

PROJECT_EPS = 1e-7


def _empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fusiondepth_amd: tensors must live on the GPU (no CPU fallback); got %s" % t.device)


# ------------------------------------------------------------------------------------ geometry ---
class _DispToDepth(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, min_depth, max_depth):
        disp = f32(disp)
        _need_cuda(disp)
        scaled, depth = torch.empty_like(disp), torch.empty_like(disp)
        call("fd_disp_to_depth_fwd", ptr(disp), ptr(scaled), ptr(depth), disp.numel(), float(min_depth),
             float(max_depth), stream())
        ctx.save_for_backward(disp)
        ctx.rng = (float(min_depth), float(max_depth))
        return scaled, depth

    @staticmethod
    def backward(ctx, g_scaled, g_depth):
        (disp,) = ctx.saved_tensors
        gs = f32(g_scaled) if g_scaled is not None else None
        gd = f32(g_depth) if g_depth is not None else None
        out = torch.empty_like(disp)
        call("fd_disp_to_depth_bwd", ptr(disp), ptr(gs), ptr(gd), ptr(out), disp.numel(), ctx.rng[0], ctx.rng[1],
             stream())
        return out, None, None


def disp_to_depth(disp, min_depth, max_depth):
    return _DispToDepth.apply(disp, min_depth, max_depth)


class _PoseMatrix(torch.autograd.Function):
    @staticmethod
    def forward(ctx, axisangle, translation, invert):
        B = axisangle.shape[0]
        aa = f32(axisangle).reshape(B, 3)
        tr = f32(translation).reshape(B, 3)
        _need_cuda(aa, tr)
        T = _empty((B, 4, 4), aa)
        call("fd_pose_matrix_fwd", ptr(aa), ptr(tr), ptr(T), B, int(bool(invert)), stream())
        ctx.save_for_backward(aa, tr)
        ctx.invert = int(bool(invert))
        ctx.shapes = (axisangle.shape, translation.shape)
        return T

    @staticmethod
    def backward(ctx, gT):
        aa, tr = ctx.saved_tensors
        B = aa.shape[0]
        gT = f32(gT)
        gaa, gtr = torch.empty_like(aa), torch.empty_like(tr)
        call("fd_pose_matrix_bwd", ptr(aa), ptr(tr), ptr(gT), ptr(gaa), ptr(gtr), B, ctx.invert, stream())
        return gaa.reshape(ctx.shapes[0]), gtr.reshape(ctx.shapes[1]), None


def transformation_from_parameters(axisangle, translation, invert=False):
    """layers.py:23-40.  axisangle / translation: [B,1,3] -> [B,4,4]."""
    return _PoseMatrix.apply(axisangle, translation, invert)


class _PoseHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pose, G, nf, Bq, invert_mask):
        import ctypes
        pose = f32(pose)
        _need_cuda(pose)
        N, ld = pose.shape
        if N != G * nf * Bq or ld % 6 != 0 or not 1 <= nf <= 4:
            raise ValueError("pose_head: pose must be [G * nf * Bq, 6 * predictions] with 1..4 frame pairs, got %s for G=%d nf=%d Bq=%d"
                             % (tuple(pose.shape), G, nf, Bq))
        Ts = [_empty((G * Bq, 4, 4), pose) for _ in range(nf)]
        aas = [_empty((G * Bq, ld // 6, 1, 3), pose) for _ in range(nf)]
        trs = [_empty((G * Bq, ld // 6, 1, 3), pose) for _ in range(nf)]
        arr = ctypes.c_void_p * nf
        call("fd_pose_head_fwd", ptr(pose), arr(*[ptr(t) for t in Ts]), arr(*[ptr(t) for t in aas]), arr(*[ptr(t) for t in trs]),
             G, nf, Bq, ld, int(invert_mask), stream())
        ctx.save_for_backward(pose)
        ctx.cfg = (G, nf, Bq, ld, int(invert_mask))
        ctx.mark_non_differentiable(*aas, *trs)
        return tuple(Ts) + tuple(aas) + tuple(trs)

    @staticmethod
    def backward(ctx, *grads):
        import ctypes
        (pose,) = ctx.saved_tensors
        G, nf, Bq, ld, invert_mask = ctx.cfg
        gTs = [None if g is None else f32(g) for g in grads[:nf]]
        g_pose = torch.empty_like(pose)
        call("fd_pose_head_bwd", ptr(pose), (ctypes.c_void_p * nf)(*[ptr(g) for g in gTs]), ptr(g_pose), G, nf, Bq, ld, invert_mask,
             stream())
        return g_pose, None, None, None, None


def pose_head(pose, groups, n_pairs, batch, inverts):
    """trainer.py:338-360 for the stacked pose network in ONE launch each way (fd_pose_head_fwd / _bwd): ``pose`` [groups * n_pairs *
    batch, 6 * predictions] = the pose decoder's output with rows ordered (micro-batch, frame pair, sample) -> per frame pair
    (cam_T_cam [groups * batch, 4, 4], axisangle, translation [groups * batch, predictions, 1, 3]).  ``inverts[k]``: trainer.py:352
    ``invert=(f_i < 0)``.  The axisangle / translation entries are what the reference's outputs dictionary holds; here they carry
    no gradient (the reference's only differentiable use of them is the matrix, except for --pose_model_type posecnn, which does not
    take this path)."""
    mask = 0
    for k, inv in enumerate(inverts):
        mask |= (1 << k) if inv else 0
    out = _PoseHead.apply(pose, int(groups), int(n_pairs), int(batch), mask)
    nf = int(n_pairs)
    return [(out[k], out[nf + k], out[2 * nf + k]) for k in range(nf)]


class _Backproject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, inv_K):
        depth, inv_K = f32(depth), f32(inv_K)
        _need_cuda(depth, inv_K)
        B, _, H, W = depth.shape
        pts = _empty((B, 4, H * W), depth)
        call("fd_backproject_fwd", ptr(depth), ptr(inv_K), ptr(pts), B, H, W, stream())
        ctx.save_for_backward(inv_K)
        ctx.shape = depth.shape
        return pts

    @staticmethod
    def backward(ctx, g):
        (inv_K,) = ctx.saved_tensors
        B, _, H, W = ctx.shape
        out = _empty(ctx.shape, g)
        call("fd_backproject_bwd", ptr(f32(g)), ptr(inv_K), ptr(out), B, H, W, stream())
        return out, None


def backproject_depth(depth, inv_K):
    return _Backproject.apply(depth, inv_K)


class _Project3D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, points, K, T, H, W, eps):
        points, K, T = f32(points), f32(K), f32(T)
        _need_cuda(points, K, T)
        B = points.shape[0]
        grid = _empty((B, H, W, 2), points)
        call("fd_project3d_fwd", ptr(points), ptr(K), ptr(T), ptr(grid), B, H, W, float(eps), stream())
        ctx.save_for_backward(points, K, T)
        ctx.dims = (B, H, W, float(eps))
        return grid

    @staticmethod
    def backward(ctx, g):
        points, K, T = ctx.saved_tensors
        B, H, W, eps = ctx.dims
        g = f32(g)
        gpts = torch.empty_like(points)
        gT = _empty((B, 4, 4), points)
        ws = _empty((query("fd_project3d_bwd_ws_floats", B, H, W),), points)
        call("fd_project3d_bwd", ptr(points), ptr(K), ptr(T), ptr(g), ptr(gpts), ptr(gT), ptr(ws), B, H, W, eps,
             stream())
        return gpts, None, gT, None, None, None


def project_3d(points, K, T, height, width, eps=PROJECT_EPS):
    return _Project3D.apply(points, K, T, height, width, eps)


def cat_xy(depth, inv_K):
    depth, inv_K = f32(depth.detach()), f32(inv_K)
    _need_cuda(depth, inv_K)
    B, _, H, W = depth.shape
    out = _empty((B, 3, H, W), depth)
    call("fd_cat_xy_fwd", ptr(depth), ptr(inv_K), ptr(out), B, H, W, stream())
    return out


class _BilinearUp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, Hout, Wout):
        x = f32(x)
        _need_cuda(x)
        B, C, Hin, Win = x.shape
        y = _empty((B, C, Hout, Wout), x)
        call("fd_bilinear_up_fwd", ptr(x), ptr(y), B * C, Hin, Win, Hout, Wout, stream())
        ctx.dims = (B, C, Hin, Win, Hout, Wout)
        return y

    @staticmethod
    def backward(ctx, g):
        B, C, Hin, Win, Hout, Wout = ctx.dims
        gx = _empty((B, C, Hin, Win), g)
        call("fd_bilinear_up_bwd", ptr(f32(g)), ptr(gx), B * C, Hin, Win, Hout, Wout, stream())
        return gx, None, None


def bilinear_upsample(x, size):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) for upsampling."""
    return _BilinearUp.apply(x, int(size[0]), int(size[1]))


# ------------------------------------------------------------------------------------ SSIM etc. ---
class _SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y):
        x, y = f32(x), f32(y)
        _need_cuda(x, y)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        call("fd_ssim_fwd", ptr(x), ptr(y), ptr(out), B, C, H, W, stream())
        ctx.save_for_backward(x, y)
        return out

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        B, C, H, W = x.shape
        gx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        gy = torch.empty_like(y) if ctx.needs_input_grad[1] else None
        call("fd_ssim_bwd", ptr(x), ptr(y), ptr(f32(g)), ptr(gx), ptr(gy), B, C, H, W, stream())
        return gx, gy


def ssim(x, y):
    return _SSIM.apply(x, y)


def reprojection_loss_map(pred, target, use_ssim=True, out=None):
    """trainer.py:476-488 without autograd (used for the identity losses): [B,3,H,W]^2 -> [B,1,H,W]."""
    pred, target = f32(pred.detach()), f32(target.detach())
    _need_cuda(pred, target)
    B, C, H, W = pred.shape
    assert C == 3
    if out is None:
        out = _empty((B, 1, H, W), pred)
        stride = H * W
    else:
        stride = out.stride(0)
    call("fd_reproj_loss_map", ptr(pred), ptr(target), out.data_ptr(), stride, B, H, W, int(bool(use_ssim)), stream())
    return out


class _SmoothLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, disp, img, normalize):
        disp, img = f32(disp), f32(img)
        _need_cuda(disp, img)
        B, _, H, W = disp.shape
        out = _empty((1,), disp)
        ws = _empty((query("fd_smooth_ws_floats", B, H, W),), disp)
        call("fd_smooth_fwd", ptr(disp), ptr(img), ptr(out), ptr(ws), B, H, W, int(normalize), stream())
        ctx.save_for_backward(disp, img)
        ctx.normalize = int(normalize)
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        disp, img = ctx.saved_tensors
        B, _, H, W = disp.shape
        g = f32(g).reshape(1)
        d = torch.empty_like(disp)
        ws = _empty((query("fd_smooth_ws_floats", B, H, W),), disp)
        call("fd_smooth_bwd", ptr(disp), ptr(img), ptr(g), ptr(d), ptr(ws), B, H, W, ctx.normalize, stream())
        return d, None, None


def get_smooth_loss(disp, img):
    """layers.py:235-248."""
    return _SmoothLoss.apply(disp, img, False)


def normalized_smooth_loss(disp, img):
    """trainer.py:569-571: get_smooth_loss(disp / (disp.mean(2,3) + 1e-7), img)."""
    return _SmoothLoss.apply(disp, img, True)


class _CombineLosses(torch.autograd.Function):
    """trainer.py:569-596 on device scalars: (loss_0..loss_{n-1}, total) = f(photo_s, smooth_s, si_s)."""

    @staticmethod
    def forward(ctx, weight, n, *terms):
        photo, smooth, si = terms[:n], terms[n:2 * n], terms[2 * n:]
        P = ctypes.c_void_p * n
        arr = [P(*[ptr(f32(t).reshape(1)) if t is not None else None for t in group]) for group in (photo, smooth, si)]
        out = _empty((n + 1,), photo[0])
        call("fd_combine_losses_fwd", ctypes.addressof(arr[0]), ctypes.addressof(arr[1]), ctypes.addressof(arr[2]), n,
             float(weight), ptr(out), stream())
        ctx.n, ctx.weight = n, float(weight)
        ctx.has = [t is not None for t in terms]
        return tuple(out[i] for i in range(n + 1))

    @staticmethod
    def backward(ctx, *gs):
        n = ctx.n
        g_total = gs[n]
        grads = _empty((3 * n,), g_total)
        call("fd_combine_losses_bwd", ptr(f32(g_total).reshape(1)), n, ctx.weight, ptr(grads), stream())
        return (None, None) + tuple(grads[i] if ctx.has[i] else None for i in range(3 * n))


def combine_losses(photo, smooth, si, smooth_weight):
    """-> ([loss_s], total) for lists of 0-dim device tensors (``si`` entries may be None).  Only ``total`` carries a
    gradient (the per-scale values are logging outputs, as in the reference)."""
    n = len(photo)
    out = _CombineLosses.apply(float(smooth_weight), n, *(list(photo) + list(smooth) + list(si)))
    return [o.detach() for o in out[:n]], out[n]


# ------------------------------------------------------------------------------------ LiDAR -------
def scatter_2channel(beam, roi=(76, 190, 2, 638), expand=2):
    """gen2channel.py:60-117 on the GPU.  beam [B,1,H,W] (or [H,W]) -> [B,2,H,W]."""
    squeeze = beam.dim() == 2
    if squeeze:
        beam = beam[None, None]
    beam = f32(beam)
    _need_cuda(beam)
    B, _, H, W = beam.shape
    out = _empty((B, 2, H, W), beam)
    call("fd_scatter_2channel", ptr(beam), ptr(out), B, H, W, roi[0], roi[1], roi[2], roi[3], expand, stream())
    return out[0] if squeeze else out


def padded_rows(im_h, target_h):
    """Rows of generate_depth_map(shape=[target_h, .]) (kitti_utils.py:88-101): top padding, 2 rows cropped if shorter."""
    return im_h + abs(target_h - im_h) - (2 if target_h < im_h else 0)


def velo_rasterize(points, P_velo2im, im_h, im_w, shape=(384, 1280), return_full=False, vel_depth=False, beam=True):
    """Velodyne scan -> "4beam" network input (kitti_utils.py:40-102 + kitti_dataset.py:93-117 + mono_dataset.py:193-198).
    ``points``: [N,4] float32 CUDA; ``P_velo2im``: 3x4 (numpy / tensor, float64); ``shape``: the reference's ``shape`` argument
    (None: no padding).  Returns the pooled float32 map (metres / 100) and / or, with ``return_full``, the float64 image
    ``generate_depth_map`` returns."""
    points = f32(points)
    _need_cuda(points)
    P = torch.as_tensor(P_velo2im, dtype=torch.float64).reshape(12).to(points.device).contiguous()
    n = points.shape[0]
    th, tw = (int(shape[0]), int(shape[1])) if shape is not None else (im_h, im_w)
    ph = padded_rows(im_h, th)
    out = torch.empty(((ph + 1) // 2, (tw + 1) // 2), device=points.device, dtype=torch.float32) if beam else None
    full = torch.empty((ph, tw), device=points.device, dtype=torch.float64) if return_full else None
    if out is None and full is None:
        raise ValueError("velo_rasterize: nothing to return")
    ws = torch.empty((query("fd_velo_rasterize_ws_bytes", n, im_h, im_w),), device=points.device, dtype=torch.uint8)
    call("fd_velo_rasterize", points.data_ptr(), n, P.data_ptr(), im_h, im_w, 1 if vel_depth else 0, th, tw,
         out.data_ptr() if out is not None else None, full.data_ptr() if full is not None else None, ws.data_ptr(), stream())
    if out is not None and full is not None:
        return out, full
    return out if out is not None else full


def scaled_roi(H, W):
    """ROI of gen2channel.py:64-65 (rows 76..189, cols 2..637 of 192x640) scaled to another size."""
    return (max(int(round(76 * H / 192)), 2), min(int(round(190 * H / 192)), H - 2), 2, W - 2)


# ------------------------------------------------------------------------------------ fused loss --
class PhotoOptions:
    """The option subset the fused loss reads (options.py:64-71,111-125,242-330)."""

    def __init__(self, min_depth=0.1, max_depth=100.0, no_ssim=False, avg_reprojection=False, si_threshold=2.0,
                 si_var=0.3, si_depth_scale=26.0, si_beam_scale=100.0, si_lo=1.0, si_mode=0):
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.no_ssim, self.avg_reprojection = bool(no_ssim), bool(avg_reprojection)
        self.si_threshold, self.si_var = float(si_threshold), float(si_var)
        self.si_depth_scale, self.si_beam_scale = float(si_depth_scale), float(si_beam_scale)
        self.si_lo = float(si_lo)
        self.si_mode = int(si_mode)       # 0 SI-log, 1 masked L1 (completor.py:718-723)


def _photo_cfg(po, B, H, W, Hs, Ws, NF, groups=1):
    c = _lib.PhotoCfg()
    c.groups = groups
    c.min_depth, c.max_depth = po.min_depth, po.max_depth
    c.B, c.H, c.W, c.Hs, c.Ws, c.NF = B, H, W, Hs, Ws, NF
    c.use_ssim = 0 if po.no_ssim else 1
    c.avg_reprojection = 1 if po.avg_reprojection else 0
    c.si_depth_scale, c.si_beam_scale = po.si_depth_scale, po.si_beam_scale
    c.si_threshold, c.si_var, c.eps = po.si_threshold, po.si_var, PROJECT_EPS
    c.si_lo = getattr(po, "si_lo", 1.0)
    c.si_mode = getattr(po, "si_mode", 0)
    return c


class _PhotoLoss(torch.autograd.Function):
    """One pyramid scale of generate_images_pred + the photometric / SI part of compute_losses (one to three source frames,
    optional predictive mask).

    Returns (to_optimise.mean(), si_loss, sel, depth, sample, color); the last four are
    non-differentiable by-products (``None`` unless requested)."""

    @staticmethod
    def forward(ctx, disp, T0, T1, T2, mask, K, inv_K, src0, src1, src2, target, ident, noise, beam, po, materialize, groups):
        disp, K, inv_K, target = f32(disp), f32(K), f32(inv_K), f32(target)
        _need_cuda(disp, K, inv_K, target, src0)
        srcs = [f32(t) for t in (src0, src1, src2) if t is not None]
        Ts = [f32(t) for t in (T0, T1, T2) if t is not None]
        NF = len(srcs)
        assert len(Ts) == NF
        B, _, Hs, Ws = disp.shape
        H, W = target.shape[2:]
        P = _empty((B, NF, 3, 4), disp)
        for f in range(NF):
            call("fd_proj_matrix_fwd", ptr(K), ptr(Ts[f]), P.data_ptr() + f * 48, NF * 12, B, stream())
        ident = f32(ident) if ident is not None else None
        noise = f32(noise) if noise is not None else None
        beam = f32(beam) if beam is not None else None
        mask = f32(mask) if mask is not None else None
        cfg = _photo_cfg(po, B, H, W, Hs, Ws, NF, groups)
        sel = _empty((B, H, W), disp, torch.uint8)
        depth = _empty((B, 1, H, W), disp) if materialize else None
        sample = _empty((NF, B, H, W, 2), disp) if materialize else None
        color = _empty((NF, B, 3, H, W), disp) if materialize else None
        reproj = _empty((B, NF, H, W), disp) if mask is not None else None
        ws = _empty((query("fd_photo_ws_floats", B, H, W),), disp)
        out = _empty((96,), disp)
        src_arr = (ctypes.c_void_p * 3)(*[ptr(srcs[min(f, NF - 1)]) for f in range(3)])
        call("fd_photo_fwd_ex", ctypes.addressof(cfg), ptr(disp), ptr(inv_K), ptr(P), ctypes.addressof(src_arr),
             ptr(target), ptr(ident), ptr(noise), ptr(beam), ptr(mask), ptr(sel), ptr(depth), ptr(sample), ptr(color),
             ptr(reproj), ptr(ws), ptr(out), stream())
        ctx.save_for_backward(disp, K, inv_K, P, target, beam, sel, out, mask, reproj, *srcs)
        ctx.cfg, ctx.NF, ctx.has_ident = cfg, NF, int(ident is not None)
        ctx.mark_non_differentiable(sel)
        extras = [t for t in (depth, sample, color) if t is not None]
        if extras:
            ctx.mark_non_differentiable(*extras)
        return out[0], out[4], sel, depth, sample, color

    @staticmethod
    def backward(ctx, g_photo, g_si, *_):
        disp, K, inv_K, P, target, beam, sel, stats, mask, reproj = ctx.saved_tensors[:10]
        srcs = ctx.saved_tensors[10:]
        cfg, NF = ctx.cfg, ctx.NF
        B, H, W = cfg.B, cfg.H, cfg.W
        g = _empty((2,), disp)
        g[0] = g_photo if g_photo is not None else 0.0
        g[1] = g_si if g_si is not None else 0.0
        d_disp = torch.empty_like(disp)
        gP = _empty((B, NF, 3, 4), disp)
        ws = _empty((query("fd_photo_bwd_ws_floats", B, H, W),), disp)
        src_arr = (ctypes.c_void_p * 3)(*[ptr(srcs[min(f, NF - 1)]) for f in range(3)])
        call("fd_photo_bwd_ex", ctypes.addressof(cfg), ptr(disp), ptr(inv_K), ptr(P), ctypes.addressof(src_arr),
             ptr(target), ptr(beam), ptr(mask), ptr(sel), ctx.has_ident, ptr(stats), ptr(g), ptr(d_disp), ptr(gP), ptr(ws),
             stream())
        gTs = []
        for f in range(3):
            if f < NF and ctx.needs_input_grad[1 + f]:
                gT = _empty((B, 4, 4), disp)
                call("fd_proj_matrix_bwd", ptr(K), gP.data_ptr() + f * 48, NF * 12, ptr(gT), B, stream())
                gTs.append(gT)
            else:
                gTs.append(None)
        g_mask = None
        if mask is not None and ctx.needs_input_grad[4]:
            # d mean(min_f mask_f r_f) / d mask_f = r_f / (B H W) where frame f won (everywhere / NF with avg_reprojection)
            if cfg.avg_reprojection and NF >= 2:
                g_mask = reproj * (g[0] / float(B * H * W * NF))
            else:
                won = sel.unsqueeze(1) == torch.arange(NF, device=sel.device, dtype=sel.dtype).view(1, NF, 1, 1)
                g_mask = reproj * won * (g[0] / float(B * H * W))
        return (d_disp, gTs[0], gTs[1], gTs[2], g_mask) + (None,) * 12


def photo_loss(disp, T_list, K, inv_K, src_list, target, ident=None, noise=None, beam=None, po=None,
               materialize=False, groups=1, mask=None):
    """Fused per-scale loss.  T_list / src_list: one to three source frames.  ``groups``: the batch is that many stacked
    micro-batches; the SI-log loss is evaluated per micro-batch and averaged.  ``mask`` [B,NF,H,W]: the predictive-mask
    baseline (trainer.py:530-541; needs ``ident is None``)."""
    po = po or PhotoOptions()
    assert 1 <= len(T_list) == len(src_list) <= 3
    Ts = list(T_list) + [None] * (3 - len(T_list))
    ss = list(src_list) + [None] * (3 - len(src_list))
    return _PhotoLoss.apply(disp, Ts[0], Ts[1], Ts[2], mask, K, inv_K, ss[0], ss[1], ss[2], target, ident, noise, beam, po,
                            materialize, int(groups))


def photo_ms_supported(po, n_src, materialize=False):
    """Configurations the multi-scale kernel (csrc/photometric_ms.hip) covers; the rest stays on the per-scale kernels."""
    return n_src == 2 and not po.no_ssim and not po.avg_reprojection and not materialize


class _PhotoLossMS(torch.autograd.Function):
    """All pyramid scales of generate_images_pred + the photometric / LiDAR part of compute_losses in ONE launch that also
    produces the unit-cotangent gradients (``fd_photo_ms_fwd``); the backward only scales them (``fd_photo_ms_bwd``).

    Returns (photo_0..photo_{S-1}, si_0..si_{S-1}, sel[S,B,H,W])."""

    @staticmethod
    def forward(ctx, T0, T1, K, inv_K, src0, src1, target, ident, noise, beam, po, groups, beam_scales, rows, *disps):
        S = len(disps)
        disps = [f32(d) for d in disps]
        K, inv_K, target, src0, src1 = f32(K), f32(inv_K), f32(target), f32(src0), f32(src1)
        _need_cuda(disps[0], K, inv_K, target, src0, src1)
        B = disps[0].shape[0]
        H, W = target.shape[2:]
        Ts = [f32(T0), f32(T1)]
        P = _empty((B, 2, 3, 4), target)
        for f in range(2):
            call("fd_proj_matrix_fwd", ptr(K), ptr(Ts[f]), P.data_ptr() + f * 48, 24, B, stream())
        ident = f32(ident) if ident is not None else None
        beam = f32(beam) if beam is not None else None
        if noise is not None and ident is not None:
            noise = [f32(n) for n in noise]            # S tensors [B,2,H,W] (or the S slices of one [S,B,2,H,W] tensor)
        else:
            noise = None
        cfg = _lib.PhotoMsCfg()
        cfg.base = _photo_cfg(po, B, H, W, H, W, 2, groups)
        cfg.n_scales = S
        for s in range(S):
            cfg.Hs[s], cfg.Ws[s] = disps[s].shape[2], disps[s].shape[3]
        cfg.beam_mask = sum(1 << s for s in beam_scales if s < S) if beam is not None else 0
        cfg.rows_per_strip = int(rows)
        need_grad = any(ctx.needs_input_grad[i] for i in (0, 1)) or any(ctx.needs_input_grad[14:])
        sel = _empty((S, B, H, W), target, torch.uint8)
        d1 = _empty((S, B, H, W), target) if need_grad else None
        ws = _empty((query("fd_photo_ms_ws_floats", ctypes.addressof(cfg)),), target)
        out = _empty((S * _lib.PHOTO_OUT_FLOATS,), target)
        PP = ctypes.c_void_p * 4
        disp_arr = PP(*[ptr(d) for d in disps])
        noise_arr = PP(*[ptr(n) for n in noise]) if noise is not None else None
        src_arr = (ctypes.c_void_p * 2)(ptr(src0), ptr(src1))
        call("fd_photo_ms_fwd", ctypes.addressof(cfg), ctypes.addressof(disp_arr), ptr(inv_K), ptr(P), ctypes.addressof(src_arr),
             ptr(target), ptr(ident), ctypes.addressof(noise_arr) if noise_arr is not None else None, ptr(beam), ptr(sel),
             ptr(d1), ptr(ws), ptr(out), stream())
        ctx.save_for_backward(K, out, ws, beam, *disps)
        ctx.d1, ctx.cfg, ctx.S = d1, cfg, S
        ctx.mark_non_differentiable(sel)
        photo = tuple(out[s * _lib.PHOTO_OUT_FLOATS] for s in range(S))
        si = tuple(out[s * _lib.PHOTO_OUT_FLOATS + 4] for s in range(S))
        return photo + si + (sel,)

    @staticmethod
    def backward(ctx, *grads):
        K, stats, ws, beam = ctx.saved_tensors[:4]
        disps = ctx.saved_tensors[4:]
        S, cfg, d1 = ctx.S, ctx.cfg, ctx.d1
        if d1 is None:
            raise RuntimeError("photo_loss_ms: backward called although no input required a gradient in forward")
        B, H, W = cfg.base.B, cfg.base.H, cfg.base.W
        PP = ctypes.c_void_p * 4

        keep = [f32(g).reshape(1) if g is not None else None for g in grads[:2 * S]]
        gp_arr = PP(*[ptr(g) for g in keep[:S]])
        gs_arr = PP(*[ptr(g) for g in keep[S:2 * S]])
        d_disps = [torch.empty_like(d) for d in disps]
        dd_arr = PP(*[ptr(d) for d in d_disps])
        disp_arr = PP(*[ptr(d) for d in disps])
        gP = _empty((B, 2, 3, 4), stats)
        call("fd_photo_ms_bwd", ctypes.addressof(cfg), ctypes.addressof(disp_arr), ptr(beam), ptr(stats), ctypes.addressof(gp_arr),
             ctypes.addressof(gs_arr), ptr(d1), ptr(ws), ctypes.addressof(dd_arr), ptr(gP), stream())
        gTs = []
        for f in range(2):
            if ctx.needs_input_grad[f]:
                gT = _empty((B, 4, 4), stats)
                call("fd_proj_matrix_bwd", ptr(K), gP.data_ptr() + f * 48, 24, ptr(gT), B, stream())
                gTs.append(gT)
            else:
                gTs.append(None)
        return (gTs[0], gTs[1]) + (None,) * 12 + tuple(d_disps)


def photo_loss_ms(disps, T_list, K, inv_K, src_list, target, ident=None, noise=None, beam=None, beam_scales=(), po=None,
                  groups=1, rows_per_strip=0):
    """Fused loss of ALL scales (two source frames).  ``noise``: S tensors [B,2,H,W] (or one [S,B,2,H,W] tensor) or None;
    ``beam_scales``: the scales that carry the LiDAR term.  Returns ([photo_s], [si_s or None], sel[S,B,H,W])."""
    po = po or PhotoOptions()
    S = len(disps)
    if not photo_ms_supported(po, len(src_list)):
        raise RuntimeError("photo_loss_ms: unsupported configuration (use photo_loss per scale)")
    res = _PhotoLossMS.apply(T_list[0], T_list[1], K, inv_K, src_list[0], src_list[1], target, ident, noise, beam, po,
                             int(groups), tuple(beam_scales), int(rows_per_strip), *disps)
    photo, si, sel = list(res[:S]), list(res[S:2 * S]), res[2 * S]
    if beam is None:
        si = [None] * S
    else:
        si = [si[s] if s in beam_scales else None for s in range(S)]
    return photo, si, sel


# ------------------------------------------------------------------------------------ conv stack --
ACT = {"none": 0, "relu": 1, "elu": 2, "sigmoid": 3, "tanh": 4}
PAD_MODE = {"zero": 0, "reflect": 1}


def _conv_desc(x, w, stride, pad, pad_mode, act, in_norm):
    d = _lib.ConvDesc()
    d.N, d.Cin, d.H, d.W = x.shape
    d.Cout, cin_w, d.KH, d.KW = w.shape
    if cin_w != d.Cin:
        raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (d.Cin, cin_w))
    d.stride, d.pad, d.pad_mode, d.act, d.in_norm = stride, pad, pad_mode, act, int(in_norm)
    return d


def _conv_out_hw(d):
    return (d.H + 2 * d.pad - d.KH) // d.stride + 1, (d.W + 2 * d.pad - d.KW) // d.stride + 1


# Cache of kernel-side weight layouts for parameters whose owner opted in (``enable_weight_cache``; the Trainer does).
# A layout is re-derived when the parameter was modified through torch (``_version``) or by the fused Adam kernel
# (``_WEIGHTS_EPOCH``, bumped by adam_step*), i.e. once per optimiser step instead of once per launch (the six ResNet
# passes of the two micro-batches reuse the same weights).  Tensors that did not opt in are re-laid-out on every call.
_WEIGHTS_EPOCH = [0]
_WT_CACHE = {}            # (cache id, kind, layout floats) -> [stamp, layout buffer, conv descriptor, weakref(parameter)]
_WT_RETIRED = []          # replaced layout buffers / job tables: a captured hipGraph may still hold their raw pointers
_NEXT_CACHE_ID = [1]


def _drop_plan():
    if _WT_PLAN[0] is not None:
        _WT_RETIRED.extend(part[0] for part in _WT_PLAN[0] if part is not None)
        _WT_PLAN[0] = None
    _sync_late_layouts(host=True)


# The re-layout launch that follows the Adam kernel is pure HBM traffic (~0.5 GB: every 3x3 weight in its forward and its
# data-gradient layout, the F(2x2, 3x3) ones at 16 floats per tap) with nothing to overlap it on the stream that just finished the
# step.  Only the small forward layouts of the first layers are needed at once; everything else ("late": data-gradient layouts,
# forward layouts from 1 MB up = ResNet layer3 / layer4 and the decoder's deep blocks) is refreshed on a side stream while the next
# step's stems and first blocks run, and a stream that is about to USE a late layout waits for that launch first
# (``_weight_layout``), as does the next optimiser step before it changes the weights again.  tuning.host.late_relayout = False: one launch.
_LATE_MIN_FLOATS = 1 << 18
_LATE = {"event": None, "waited": set(), "stream": None}


def _late_relayout_on():
    return tuning.host.late_relayout and not torch.cuda.is_current_stream_capturing()


def _wait_late_layouts():
    """Order the current stream behind the pending side-stream re-layout (once per stream and optimiser step)."""
    ev = _LATE["event"]
    if ev is None:
        return
    sid = stream()
    if sid not in _LATE["waited"]:
        torch.cuda.current_stream().wait_event(ev)
        _LATE["waited"].add(sid)


def sync_late_layouts():
    """Order the current stream behind a pending side-stream re-layout and forget it (call before capturing a hipGraph, before
    touching the cached layouts from outside)."""
    _sync_late_layouts()


def _sync_late_layouts(host=False):
    """Forget the pending late re-layout after ordering the current stream behind it.  ``host``: wait on the host instead - for
    callers in the middle of a step (a new layout shape dropped the plan), where other streams may be about to use a late layout
    without passing through the current stream first."""
    if _LATE["event"] is not None:
        if host and not torch.cuda.is_current_stream_capturing():
            _LATE["event"].synchronize()
        torch.cuda.current_stream().wait_event(_LATE["event"])
        _LATE["event"] = None
        _LATE["waited"] = set()


def evict_dead_weight_layouts():
    """Forget the layouts of parameters that no longer exist (a Trainer that was deleted): their buffers move to the retired
    list instead of being freed at once, because a hipGraph captured by that Trainer may still reference them; returns the
    number of entries dropped.  ``release_retired_layouts()`` frees the list once no such graph can be replayed any more."""
    dead = [k for k, e in _WT_CACHE.items() if e[3]() is None]
    for k in dead:
        _WT_RETIRED.append(_WT_CACHE.pop(k)[1])
    if dead:
        _drop_plan()
    return len(dead)


def release_retired_layouts():
    n = len(_WT_RETIRED)
    _WT_RETIRED.clear()
    return n


def bump_weights_epoch():
    _WEIGHTS_EPOCH[0] += 1


_FROZEN_EPOCH = [0]


def invalidate_frozen_layouts():
    """Frozen weights' layouts ignore the per-optimiser-step epoch; their stamp is (``_version``, this epoch, ``data_ptr``).  A write that
    torch's version counter does not see - ``p.data.copy_``, a broadcast into ``p.data``, a raw kernel - must be followed by this call
    (``Refiner._load_pretrained``, ``Trainer.load_model`` and ``dp.broadcast_module_state`` do it): every frozen layout is re-derived
    at its next use."""
    _FROZEN_EPOCH[0] += 1


def enable_weight_cache(params, frozen=False):
    """``frozen``: weights that no optimiser touches (the Refiner's stage-1 networks).  Their layouts are derived once and stay valid
    across optimiser steps (the epoch stamp that invalidates trained weights' layouts after every Adam launch is ignored; an in-place
    change through torch - ``load_state_dict`` - still bumps ``_version``), and they stay out of the post-Adam refresh launch.
    Round 4 re-laid them out on EVERY call: 100 launches per Refiner step."""
    for p in params:
        if p.dim() == 4 and not hasattr(p, "_fd_cache_id"):
            p._fd_cache_id = _NEXT_CACHE_ID[0]
            _NEXT_CACHE_ID[0] += 1
        if p.dim() == 4:
            if frozen:
                p._fd_frozen = True
            elif getattr(p, "_fd_frozen", False):          # a formerly frozen parameter that is trained now: back under the optimiser epoch
                p._fd_frozen = False
                _drop_plan()


def _layout_stamp(w):
    if getattr(w, "_fd_frozen", False):
        return (w._version, -1 - _FROZEN_EPOCH[0], w.data_ptr())
    return (w._version, _WEIGHTS_EPOCH[0], w.data_ptr())


def enable_direct_grad(params):
    """Opt-in: weight / bias / BatchNorm-affine gradients are accumulated by the backward kernels straight into the
    pre-allocated ``param.grad`` (a view of the trainer's flat gradient buffer) instead of being returned to autograd,
    which would launch one ATen add per parameter per micro-batch (~600 tiny kernels per optimiser step)."""
    for p in params:
        p._fd_direct_grad = True
        if getattr(p, "_fd_frozen", False):                # it gets gradients, so something will change it
            p._fd_frozen = False
            _drop_plan()


# ---- "this parameter's gradient is complete" notifications ------------------------------------------------------------------
# With in-place accumulation autograd never sees a parameter gradient, so post-accumulate hooks do not fire.  The backward
# wrappers call this right after LAUNCHING the kernel that finishes a parameter's gradient; dp.GradientSynchronizer uses it to
# issue a bucket's all-reduce behind that kernel on the same stream while the rest of the backward pass is still being issued.
_GRAD_READY = []        # weak references to the bound callbacks of live subscribers (one per GradientSynchronizer with world > 1)
_PARAM_USES = {}        # id(param) -> number of forward uses since begin_forward_pass() (a network may run twice per pass)


def add_grad_ready_callback(bound_method):
    """Subscribe ``bound_method(param)``.  Held weakly: a deleted trainer's synchroniser (and its flat buffers) is not kept
    alive by this module, and several trainers in one process (a Trainer and a Refiner, two Trainers) each keep their overlap -
    every subscriber is told about every parameter and ignores the ones it does not own."""
    _GRAD_READY.append(weakref.WeakMethod(bound_method))


def _live_grad_ready():
    live = [(r, r()) for r in _GRAD_READY]
    if any(cb is None for _, cb in live):
        _GRAD_READY[:] = [r for r, cb in live if cb is not None]
    return [cb for _, cb in live if cb is not None]


def begin_forward_pass():
    """Start counting parameter uses afresh: the backward pass of this forward runs one gradient kernel per use.  Side-stream
    weight gradients of a previous backward pass that nobody joined (a caller driving process_batch + backward itself, without
    Trainer._join_side_streams) are joined here, so that the tensors they keep alive are released at the latest one pass later."""
    _PARAM_USES.clear()
    if _WGRAD_KEEPALIVE:
        join_wgrad_streams()


def param_uses(p):
    return _PARAM_USES.get(id(p), 1)


def _note_use(*params):
    if _lib.RECORDER[0] is not None:
        _lib.RECORDER[0].side("note_use", params)
    if _GRAD_READY:
        for p in params:
            if p is not None:
                _PARAM_USES[id(p)] = _PARAM_USES.get(id(p), 0) + 1


def _grad_ready(*params):
    rec = _lib.RECORDER[0]
    if rec is not None:
        rec.side("grad_ready", params)
        if rec.mute_grad_ready:
            return                      # an isolated recording pass: its gradients are thrown away, nobody may be told
    if _GRAD_READY:
        for cb in _live_grad_ready():
            for p in params:
                if p is not None:
                    cb(p)


def _direct_grad_target(p):
    if p is not None and getattr(p, "_fd_direct_grad", False) and p.grad is not None and p.grad.is_contiguous():
        return p.grad
    return None


def _weight_layout(w, cache_id, kind, nfloats, desc=None):
    """-> (buffer, ready flag) for weight ``w`` and layout ``kind`` ('f' forward, 'd' data-gradient)."""
    if cache_id is None:
        return torch.empty((nfloats,), device=w.device, dtype=torch.float32), 0
    # nfloats is part of the key: the layout FORMAT of a weight depends on the kernel family its input shape is routed to (12 or 16
    # floats per tap: conv_wino.hip::wino_fwd_mode looks at N*H*W), so a weight used at two batch sizes (the stacked training batch and
    # a smaller validation batch) keeps both layouts resident instead of rebuilding one over the other on every switch
    key = (cache_id, kind, nfloats)
    stamp = _layout_stamp(w)
    ent = _WT_CACHE.get(key)
    if _lib.RECORDER[0] is not None:
        _lib.RECORDER[0].side("layout", (w, key))
    if ent is not None:
        if _LATE["event"] is not None and ent[4]:
            _wait_late_layouts()
        if ent[6] and not getattr(w, "_fd_frozen", False) and _WEIGHTS_EPOCH[0] - ent[5] > _PLAN_IDLE_STEPS:
            _drop_plan()                # it left the per-step refresh while idle and is in use again: let the next plan include it
        ent[5] = _WEIGHTS_EPOCH[0]
        if ent[0] == stamp:
            return ent[1], 1
        ent[0] = stamp
        return ent[1], 0
    buf = torch.empty((nfloats,), device=w.device, dtype=torch.float32)
    # [stamp, layout buffer, conv descriptor, weakref(parameter), refreshed on the side stream, epoch of the last use, left out of the plan]
    _WT_CACHE[key] = [stamp, buf, desc, weakref.ref(w), False, _WEIGHTS_EPOCH[0], False]
    if not getattr(w, "_fd_frozen", False):
        _drop_plan()                    # a layout the plan does not know: fall back to per-call re-layout until rebuilt
    return buf, 0


# One-launch refresh of every cached layout (fd_relayout_batch): built once the cache is populated (after the first full
# forward + backward), run by adam_step* right after the weights change.
# A trained weight keeps one layout per input shape it was used at (the stacked training batch AND the validation batch: the format
# depends on the kernel family).  A layout that no convolution has used for _PLAN_IDLE_STEPS optimiser steps stays OUT of the per-step
# refresh - it is re-derived lazily by its next user through the stamp mismatch, which also drops the plan so that the layout is
# refreshed with the others again while it stays in use (a validation pass of several batches: one rebuild, then one launch per step) -
# and after _RETIRE_IDLE_STEPS its buffer is released (a rare shape no longer costs traffic and memory on every step).
_WT_PLAN = [None]
_PLAN_IDLE_STEPS = 8
_RETIRE_IDLE_STEPS = 512


def build_weight_plan():
    """Collect the re-layout jobs of every cached weight layout into a device table; returns the number of jobs."""
    from ._lib import RelayoutJob
    evict_dead_weight_layouts()
    _PLAN_STALE[0] = False
    now = _WEIGHTS_EPOCH[0]
    for k in [k for k, e in _WT_CACHE.items() if now - e[5] > _RETIRE_IDLE_STEPS and not getattr(e[3](), "_fd_frozen", False)]:
        _WT_RETIRED.append(_WT_CACHE.pop(k)[1])
    ents = []
    for k, e in _WT_CACHE.items():
        if e[2] is None or getattr(e[3](), "_fd_frozen", False):
            continue
        e[6] = now - e[5] > _PLAN_IDLE_STEPS
        if not e[6]:
            ents.append((k, e))
    if not ents:
        _drop_plan()
        return 0
    def table(part):
        if not part:
            return None, 0
        jobs = (RelayoutJob * (4 * len(part)))()
        n = 0
        for (cid, kind, _nf), e in part:
            n += query("fd_conv2d_relayout_jobs", ctypes.addressof(e[2]), 0 if kind == "f" else 1, ptr(e[3]()), ptr(e[1]),
                       ctypes.addressof(jobs) + n * ctypes.sizeof(RelayoutJob))
        if n == 0:
            return None, 0
        blocks = query("fd_relayout_plan", ctypes.addressof(jobs), n)
        raw = bytes(memoryview(jobs))[: n * ctypes.sizeof(RelayoutJob)]
        dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(part[0][1][1].device)
        return (dev, n, blocks, [k for k, _ in part]), n
    is_late = lambda k, e: k[1] != "f" or e[1].numel() >= _LATE_MIN_FLOATS
    for k, e in ents:
        e[4] = bool(is_late(k, e))
    early, n_early = table([(k, e) for k, e in ents if not e[4]])
    late, n_late = table([(k, e) for k, e in ents if e[4]])
    _drop_plan()
    if n_early + n_late == 0:
        return 0
    _WT_PLAN[0] = (early, late)
    return n_early + n_late


def refresh_weight_layouts():
    """Re-derive every planned layout from the current weights (one launch) and mark them valid for this epoch."""
    plan = _WT_PLAN[0]
    if plan is None:
        return False
    early, late = plan
    _sync_late_layouts()                       # (a refresh without an optimiser step in between: never two in flight)
    for part in (early, late):
        if part is None:
            continue
        dev, n, blocks, keys = part
        if part is late and _late_relayout_on():
            cur = torch.cuda.current_stream()
            if _LATE["stream"] is None:
                _LATE["stream"] = torch.cuda.Stream()
            side = _LATE["stream"]
            side.wait_stream(cur)              # behind the Adam kernel (and everything that read the old layouts)
            with torch.cuda.stream(side):
                call("fd_relayout_batch", ptr(dev), n, blocks, stream())
                ev = torch.cuda.Event()
                ev.record(side)
            _LATE["event"], _LATE["waited"] = ev, set()
        else:
            call("fd_relayout_batch", ptr(dev), n, blocks, stream())
        for k in keys:
            e = _WT_CACHE[k]
            w = e[3]()
            if w is not None:
                e[0] = _layout_stamp(w)
            if _WEIGHTS_EPOCH[0] - e[5] > _PLAN_IDLE_STEPS:
                _PLAN_STALE[0] = True          # refreshed, but nobody has read it for a while: the owner rebuilds the plan without it
    return True


_PLAN_STALE = [False]


def weight_plan_needs_rebuild():
    """No plan yet, or the plan carries layouts that have gone idle (``_PLAN_IDLE_STEPS``): ``build_weight_plan`` leaves those out."""
    return _WT_PLAN[0] is None or _PLAN_STALE[0]


# Shape-keyed plans: the descriptor of a convolution and the workspace / layout sizes the library reports for it depend only on
# the shapes and flags, so they are asked for once per distinct layer shape instead of on every launch (a step launches ~220
# convolutions forward and as many backward; the size queries alone were ~1 300 library calls per step).
_CONV_PLANS = {}
_CONV_PLANS_GEN = [0]
_PLANS_EVER = []          # every descriptor ever created stays allocated (a few hundred bytes each; recorded call sequences hold their addresses)


class _ConvPlan:
    __slots__ = ("d", "dp", "Ho", "Wo", "fwd_ws", "fwd_wt", "bwd_data_ws", "bwd_data_wt", "bwd_weight_ws", "stat_slots", "bn_fused")

    def __init__(self, x, w, stride, pad, pad_mode, act, in_norm):
        self.d = _conv_desc(x, w, stride, pad, pad_mode, act, in_norm)
        self.dp = ctypes.addressof(self.d)
        _lib.HOST_PERSISTENT.add(self.dp)          # may appear as a literal in a recorded call sequence (replay.py)
        _PLANS_EVER.append(self)                   # ... so the descriptor must outlive a cleared plan table
        self.Ho, self.Wo = _conv_out_hw(self.d)
        self.fwd_ws = query("fd_conv2d_fwd_ws_floats", self.dp)
        self.fwd_wt = query("fd_conv2d_fwd_wt_floats", self.dp)
        self.stat_slots = query("fd_conv2d_fwd_stat_slots", self.dp)       # BatchNorm partial sums per (image, channel); 0: none
        self.bwd_data_ws = self.bwd_data_wt = self.bwd_weight_ws = None
        self.bn_fused = {}                 # BatchNorm groups -> fd_conv2d_fwd_bn_ok

    def fused_bn_ok(self, groups):
        ok = self.bn_fused.get(groups)
        if ok is None:
            ok = self.bn_fused[groups] = bool(query("fd_conv2d_fwd_bn_ok", self.dp, int(groups)))
        return ok

    def data_sizes(self):
        if self.bwd_data_ws is None:
            self.bwd_data_ws = max(query("fd_conv2d_bwd_data_ws_floats", self.dp), 1)
            self.bwd_data_wt = query("fd_conv2d_bwd_data_wt_floats", self.dp)
        return self.bwd_data_ws, self.bwd_data_wt

    def weight_ws(self):
        if self.bwd_weight_ws is None:
            self.bwd_weight_ws = max(query("fd_conv2d_bwd_weight_ws_floats", self.dp), 1)
        return self.bwd_weight_ws


# The library's kernel-selection thresholds (fd_tuning) change the split-K / slab workspace and weight-layout sizes, so the number of
# fd_set_tuning calls so far is part of the plan key (one integer compare; the tests and sweeps flip thresholds within a process).
def _conv_plan(x, w, stride, pad, pad_mode, act, in_norm):
    gen = tuning.generation()
    if gen != _CONV_PLANS_GEN[0]:          # plans of an older fd_tuning can never be hit again: drop them (sweeps flip thresholds in loops)
        _CONV_PLANS.clear()
        _CONV_PLANS_GEN[0] = gen
    key = (tuple(x.shape), tuple(w.shape), stride, pad, pad_mode, act, in_norm)
    plan = _CONV_PLANS.get(key)
    if plan is None:
        plan = _CONV_PLANS[key] = _ConvPlan(x, w, stride, pad, pad_mode, act, in_norm)
    return plan


# direct-equivalent convolution flops issued since the tally was switched on (bench.py: the MFMA fraction of the configurations that
# have no analytic table - Refiner / Completor steps); None = off, [0.0] = counting
CONV_FLOP_TALLY = None


def _tally(d, Ho, Wo, passes=1):
    if CONV_FLOP_TALLY is not None:
        CONV_FLOP_TALLY[0] += 2.0 * passes * d.N * d.Cout * Ho * Wo * d.Cin * d.KH * d.KW


def _conv_forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm, want_stats=False):
    """-> (x as float32, y, part): ``part`` [N, Cout, S, 2] holds the per-channel (sum, M2 about the slot's own mean) of y per pixel slot,
    gathered in the convolution's epilogue for the BatchNorm that follows - or is None when ``want_stats`` is False or the
    kernel chosen for this shape has no statistics epilogue."""
    cache_id = getattr(w, "_fd_cache_id", None)
    ctx.params = (w, bias)
    _note_use(w, bias)
    x, w = f32(x), f32(w)
    bias = f32(bias) if bias is not None else None
    _need_cuda(x, w)
    plan = _conv_plan(x, w, stride, pad, pad_mode, act, bool(in_norm))
    d, Ho, Wo, nws, nwt = plan.d, plan.Ho, plan.Wo, plan.fwd_ws, plan.fwd_wt
    y = _empty((d.N, d.Cout, Ho, Wo), x)
    _tally(d, Ho, Wo)
    ws = _empty((nws,), x) if nws > 0 else None
    wt, ready = _weight_layout(w, cache_id, "f", nwt, d) if nwt > 0 else (None, 0)
    part = None
    if want_stats and plan.stat_slots > 0:
        part = _empty((d.N, d.Cout, plan.stat_slots, 2), x)
        call("fd_conv2d_fwd_stats", plan.dp, ptr(x), ptr(w), ptr(bias), ptr(y), ptr(wt), ready, ptr(ws), ptr(part), stream())
    else:
        call("fd_conv2d_fwd", plan.dp, ptr(x), ptr(w), ptr(bias), ptr(y), ptr(wt), ready, ptr(ws), stream())
    ctx.save_for_backward(x, w, y if act != 0 else None)
    ctx.desc, ctx.has_bias, ctx.cache_id, ctx.plan = d, bias is not None, cache_id, plan
    return x, y, part


def _conv_backward(ctx, gy, gx_add=None):
    """-> (gx, gw, gb).  ``gx_add``: a second gradient arriving at the conv's input (see ``conv2d_tap``); it joins the data
    gradient in the kernel's epilogue (fd_conv2d_bwd_data_add)."""
    x, w, y = ctx.saved_tensors
    d = ctx.desc
    gx = gw = gb = None
    if gy is None:                   # only the tap was used downstream
        return (f32(gx_add) if gx_add is not None else None), None, None
    gy = f32(gy)
    if d.act != 0 and not getattr(ctx, "grad_preact", False):
        gpre = torch.empty_like(gy)
        call("fd_act_bwd", ptr(y), ptr(gy), ptr(gpre), gy.numel(), d.act, stream())
        gy = gpre
    plan = ctx.plan
    dp = plan.dp
    in_act = getattr(ctx, "in_act", 0)
    if in_act and ctx.needs_input_grad[0] and (gx_add is not None or d.in_norm):
        # the producer of x was built with grad_preact=True and has skipped its own act' pass: this layer MUST multiply by act'(x)
        # (the fallback branches below would drop it silently)
        raise RuntimeError("conv2d(in_act=...): the fused act' data gradient cannot be taken here (second incoming gradient or "
                           "normalised input) - build the producer without grad_preact")
    if ctx.needs_input_grad[0]:
        _tally(d, plan.Ho, plan.Wo)
        gx = torch.empty_like(x)
        n_ws, n_wt = plan.data_sizes()
        ws = _empty((n_ws,), x)
        wt, ready = _weight_layout(w, ctx.cache_id, "d", n_wt, d)
        if in_act:
            # x is an activation output consumed by this layer alone: hand its producer the gradient w.r.t. the PRE-activation
            call("fd_conv2d_bwd_data_inact", dp, ptr(gy), ptr(w), ptr(x), in_act, ptr(gx), ptr(wt), ready, ptr(ws), stream())
        elif gx_add is not None and not d.in_norm:
            call("fd_conv2d_bwd_data_add", dp, ptr(gy), ptr(w), ptr(f32(gx_add)), ptr(gx), ptr(wt), ready, ptr(ws), stream())
        else:
            call("fd_conv2d_bwd_data", dp, ptr(gy), ptr(w), ptr(gx), ptr(wt), ready, ptr(ws), stream())
            if d.in_norm:   # d/dx of (x - 0.45) / 0.225
                call("fd_axpby", ptr(gx), ptr(gx), ptr(gx), gx.numel(), 1.0 / 0.225, 0.0, stream())
            if gx_add is not None:
                call("fd_axpby", ptr(gx), ptr(f32(gx_add)), ptr(gx), gx.numel(), 1.0, 1.0, stream())
    if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
        _tally(d, plan.Ho, plan.Wo)
        tw = _direct_grad_target(ctx.params[0])
        tb = _direct_grad_target(ctx.params[1]) if ctx.has_bias else None
        direct = tw is not None and (not ctx.has_bias or tb is not None)
        gw = tw if direct else torch.empty_like(w)
        gb = (tb if direct else _empty((d.Cout,), x)) if ctx.has_bias else None
        ws = _empty((plan.weight_ws(),), x)
        if direct and getattr(ctx.params[0], "_fd_side_wgrad", False):
            # a layer of the decoder's serial chain (enable_side_wgrad): its weight gradient is a leaf of the backward graph - it
            # runs on a side stream beside the data gradients of the following layers; the gradient-ready notification is given
            # with that stream current, so that a data-parallel bucket is ordered behind the kernel that really finishes it
            side = _wgrad_stream()                     # ordered after everything queued so far (gy is complete)
            with torch.cuda.stream(side):
                call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws), 1, stream())
                _grad_ready(ctx.params[0], ctx.params[1] if ctx.has_bias else None)
            _WGRAD_KEEPALIVE.append((x, gy, ws))
            return gx, None, None
        call("fd_conv2d_bwd_weight", dp, ptr(x), ptr(gy), ptr(gw), ptr(gb), ptr(ws), int(direct), stream())
        if direct:
            gw = gb = None          # already accumulated in place
            _grad_ready(ctx.params[0], ctx.params[1] if ctx.has_bias else None)
    return gx, gw, gb


# ---- weight gradients of the decoder on a side stream -----------------------------------------------------------------------
# Decoder -> loss -> decoder is the serial section of the step: one kernel at a time on the main stream, at batch 12 and 16-128
# channels, while the encoder streams have little or nothing to run.  In a conv's backward only the data gradient feeds the next
# layer; the weight gradient (+ its slab reduction + the bias sums) is a leaf, so for parameters marked by ``enable_side_wgrad`` it
# is issued on ONE side stream per issuing stream.  The tensors those kernels read are kept alive until ``join_wgrad_streams``
# (the caching allocator would otherwise hand their memory to later kernels of the issuing stream).  Round 2's FD_ASYNC_WGRAD did
# this for EVERY convolution - slower (the encoders' streams already fill the chip) and its notifications were given on the wrong
# stream; this is the decoder only, opt-in per parameter.
_WGRAD_STREAMS = {}
_WGRAD_KEEPALIVE = []


def enable_side_wgrad(params, on=True):
    for p in params:
        if p.dim() == 4:
            p._fd_side_wgrad = bool(on)


def join_wgrad_streams():
    """Order every side-stream weight gradient before what follows on the current stream (optimiser / all-reduce)."""
    if not _WGRAD_STREAMS:
        return
    cur = torch.cuda.current_stream()
    for st in _WGRAD_STREAMS.values():
        cur.wait_stream(st)
    _WGRAD_KEEPALIVE.clear()


def _wgrad_stream():
    cur = torch.cuda.current_stream()
    st = _WGRAD_STREAMS.get(cur.cuda_stream)
    if st is None:
        st = _WGRAD_STREAMS[cur.cuda_stream] = torch.cuda.Stream()
    st.wait_stream(cur)
    return st


class _Conv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm, grad_preact=False, in_act=0):
        ctx.grad_preact, ctx.in_act = bool(grad_preact), int(in_act)
        return _conv_forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm)[1]

    @staticmethod
    def backward(ctx, gy):
        return _conv_backward(ctx, gy) + (None, None, None, None, None, None, None)


_NO_STATS = {}


def _no_stats(like):
    """Placeholder for "this convolution has no statistics epilogue" (autograd Functions return tensors)."""
    t = _NO_STATS.get(like.device)
    if t is None:
        t = _NO_STATS[like.device] = torch.empty((0,), device=like.device)
    return t


class _Conv2dStats(torch.autograd.Function):
    """conv2d that also returns the BatchNorm partial sums of its output (``fd_conv2d_fwd_stats``); an empty tensor when the
    kernel chosen for the shape cannot produce them."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm):
        _, y, part = _conv_forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm, True)
        part = part if part is not None else _no_stats(y)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)      # no zero-filled "gradient" of `part` (one fill launch per convolution otherwise)
        return y, part

    @staticmethod
    def backward(ctx, gy, _g_part):
        return _conv_backward(ctx, gy) + (None, None, None, None, None)


class _Conv2dTapStats(torch.autograd.Function):
    """``_Conv2dTap`` + the BatchNorm partial sums of ``_Conv2dStats``."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm):
        xf, y, part = _conv_forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm, True)
        part = part if part is not None else _no_stats(y)
        ctx.mark_non_differentiable(part)
        ctx.set_materialize_grads(False)
        return y, xf.view_as(xf), part

    @staticmethod
    def backward(ctx, gy, g_tap, _g_part):
        return _conv_backward(ctx, gy, g_tap) + (None, None, None, None, None)


def conv2d_stats(x, weight, bias=None, stride=1, pad=0, pad_mode="zero", tap=False):
    """Convolution followed by a training-mode BatchNorm: -> (y, conv_stats[, x_tap]) where ``conv_stats`` goes to
    ``batch_norm(..., conv_stats=)`` (None when this shape's kernel cannot gather them: batch_norm then makes its own pass)."""
    args = (x, weight, bias, int(stride), int(pad), PAD_MODE[pad_mode], 0, False)
    if tap and torch.is_grad_enabled() and x.requires_grad:
        y, x_tap, part = _Conv2dTapStats.apply(*args)
    else:
        y, part = _Conv2dStats.apply(*args)
        x_tap = x
    part = part if part.numel() else None
    return (y, part, x_tap) if tap else (y, part)


class _Conv2dTap(torch.autograd.Function):
    """conv2d that also hands its input on: (y, x_tap) with x_tap == x.  A tensor that feeds a convolution and something else (the
    input of a ResNet block is also its residual branch) normally receives two gradients that autograd sums with an element-wise
    kernel; routing the second consumer through ``x_tap`` brings that gradient to THIS node instead, where it is added in the
    epilogue of the data-gradient kernel.  Same values, one pass over the tensor less, one launch less."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm):
        xf, y, _ = _conv_forward(ctx, x, w, bias, stride, pad, pad_mode, act, in_norm)
        return y, xf.view_as(xf)

    @staticmethod
    def backward(ctx, gy, g_tap):
        return _conv_backward(ctx, gy, g_tap) + (None, None, None, None, None)


def conv2d_tap(x, weight, bias=None, stride=1, pad=0, pad_mode="zero", act="none", in_norm=False):
    """-> (conv2d(x, ...), x): use the second value wherever ``x`` itself is needed again (see ``_Conv2dTap``)."""
    if not (torch.is_grad_enabled() and x.requires_grad):
        return conv2d(x, weight, bias, stride, pad, pad_mode, act, in_norm), x
    return _Conv2dTap.apply(x, weight, bias, int(stride), int(pad), PAD_MODE[pad_mode], ACT[act], bool(in_norm))


class _InputNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean, std):
        x = f32(x)
        _need_cuda(x)
        y = torch.empty_like(x)
        call("fd_input_normalize", ptr(x), ptr(y), x.numel(), float(mean), float(std), stream())
        ctx.std = float(std)
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = f32(gy)
        gx = torch.empty_like(gy)
        call("fd_axpby", ptr(gy), ptr(gy), ptr(gx), gy.numel(), 1.0 / ctx.std, 0.0, stream())
        return gx, None, None


def stack_normalize(pieces, n_images, channels_total, normalize=True, mean=0.45, std=0.225):
    """One launch for ``torch.cat`` along batch and channels + the encoder's input normalisation: ``pieces`` = [(tensor [imgs, C, H, W]
    contiguous, first destination image, destination channel offset)]; all pieces have the same shape.  -> [n_images, channels_total,
    H, W] (fd_stack_normalize).  No gradient (network inputs)."""
    t0 = f32(pieces[0][0].detach())
    _need_cuda(t0)
    imgs, C, H, W = t0.shape
    out = _empty((n_images, channels_total, H, W), t0)
    for lo in range(0, len(pieces), 16):
        part = pieces[lo:lo + 16]
        ts = [f32(t.detach()) for t, _, _ in part]
        if any(t.shape != t0.shape for t in ts):
            raise RuntimeError("stack_normalize: pieces differ in shape")
        n = len(part)
        src = (ctypes.c_void_p * n)(*[ptr(t) for t in ts])
        di = (ctypes.c_int * n)(*[int(d) for _, d, _ in part])
        dc = (ctypes.c_int * n)(*[int(c) for _, _, c in part])
        call("fd_stack_normalize", src, di, dc, n, imgs, C, channels_total, H, W, ptr(out), int(bool(normalize)), float(mean), float(std), stream())
    return out


def input_normalize(x, mean=0.45, std=0.225):
    """``(x - mean) / std`` — the encoder's input normalisation (resnet_encoder.py:94) as its own pass, so that the stem
    convolution and its weight gradient gather plain values (the fused ``in_norm`` variant pays a division per tap)."""
    return _InputNormalize.apply(x, mean, std)


def conv2d(x, weight, bias=None, stride=1, pad=0, pad_mode="zero", act="none", in_norm=False, grad_preact=False, in_act="none"):
    """act(conv2d(pad(x)) + bias) on the MFMA implicit-GEMM kernels; ``in_norm`` folds the encoder's
    (x-0.45)/0.225 into the tap loads (resnet_encoder.py:94).

    A contract between a layer with an activation and the ONE consumer of its output, so that ``act'`` is applied where the
    gradient is produced instead of in an element-wise pass of its own (the depth decoder sets both ends, networks/depth_decoder.py):
    ``grad_preact``: the gradient this call receives already is the gradient w.r.t. its pre-activation (its consumer multiplied by
    act'(y) - ``upsample_concat(..., a_act=)`` or a ``conv2d(..., in_act=)``); ``in_act``: ``x`` is the output of that activation
    and feeds this call alone - the data gradient returned is multiplied by act'(x)."""
    if (grad_preact or ACT[in_act]) and torch.is_grad_enabled():
        return _Conv2d.apply(x, weight, bias, int(stride), int(pad), PAD_MODE[pad_mode], ACT[act], bool(in_norm), bool(grad_preact), ACT[in_act])
    return _Conv2d.apply(x, weight, bias, int(stride), int(pad), PAD_MODE[pad_mode], ACT[act], bool(in_norm))


def _bn_forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu, groups, conv_stats=None):
    """Body of ``_BatchNorm.forward``; ``ctx`` is the Function's context or the BatchNorm half of a fused node (``_Part``)."""
    ctx.params = (weight, bias)
    _note_use(weight, bias)
    ctx.groups = groups
    x = f32(x)
    _need_cuda(x)
    N, C, H, W = x.shape
    y = torch.empty_like(x)
    res = f32(residual) if residual is not None else None
    if training:
        mean, invstd = _empty((groups * C,), x), _empty((groups * C,), x)
        if conv_stats is not None and groups <= 16:
            # statistics gathered by the producing convolution's epilogue: one launch, no statistics pass over x
            assert conv_stats.shape[:2] == (N, C) and conv_stats.shape[3] == 2
            call("fd_bn_train_fwd_parts", ptr(x), ptr(weight), ptr(bias), ptr(res), ptr(y), ptr(running_mean), ptr(running_var),
                 ptr(mean), ptr(invstd), ptr(conv_stats), int(conv_stats.shape[2]), N, C, H, W, groups, float(eps),
                 float(momentum), int(relu), stream())
        else:
            ws = _empty((query("fd_bn_ws_floats", N, C, H, W, groups),), x)
            call("fd_bn_train_fwd", ptr(x), ptr(weight), ptr(bias), ptr(res), ptr(y), ptr(running_mean), ptr(running_var),
                 ptr(mean), ptr(invstd), ptr(ws), N, C, H, W, groups, float(eps), float(momentum), int(relu), stream())
        # ReLU without a residual input: the backward recomputes the mask from x (fd_bn_train_bwd_remask) and does not read y
        ctx.remask = bool(relu and residual is None and tuning.host.bn_remask)
        ctx.save_for_backward(x, (bias if ctx.remask else y) if relu else None, weight, mean, invstd)
    else:
        call("fd_bn_eval_fwd", ptr(x), ptr(weight), ptr(bias), ptr(res), ptr(y), ptr(running_mean), ptr(running_var),
             N, C, H, W, float(eps), int(relu), stream())
    ctx.training, ctx.relu, ctx.has_res = bool(training), int(relu), residual is not None
    return y


def _bn_backward(ctx, gy, need_res):
    """Body of ``_BatchNorm.backward`` -> (gx, gweight, gbias, gresidual)."""
    if not ctx.training:
        raise RuntimeError("BatchNorm backward is implemented for training mode only (frozen nets run under no_grad)")
    x, y, weight, mean, invstd = ctx.saved_tensors
    N, C, H, W = x.shape
    gy = f32(gy)
    gx = torch.empty_like(x)
    tw, tb = _direct_grad_target(ctx.params[0]), _direct_grad_target(ctx.params[1])
    direct = tw is not None and tb is not None
    gw, gb = (tw, tb) if direct else (_empty((C,), x), _empty((C,), x))
    gres = torch.empty_like(x) if ctx.has_res and need_res else None
    ws = _empty((query("fd_bn_ws_floats", N, C, H, W, ctx.groups),), x)
    if getattr(ctx, "remask", False):             # `y` holds the bias here (see _bn_forward)
        call("fd_bn_train_bwd_remask", ptr(x), ptr(gy), ptr(weight), ptr(y), ptr(mean), ptr(invstd), ptr(gx), ptr(gw), ptr(gb),
             ptr(ws), N, C, H, W, ctx.groups, int(direct), stream())
    else:
        call("fd_bn_train_bwd", ptr(x), ptr(y), ptr(gy), ptr(weight), ptr(mean), ptr(invstd), ptr(gx), ptr(gw), ptr(gb),
             ptr(gres), ptr(ws), N, C, H, W, ctx.groups, ctx.relu, int(direct), stream())
    if direct:
        gw = gb = None
        _grad_ready(ctx.params[0], ctx.params[1])
    return gx, gw, gb, gres


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu, groups, conv_stats=None):
        return _bn_forward(ctx, x, weight, bias, residual, running_mean, running_var, training, momentum, eps, relu, groups, conv_stats)

    @staticmethod
    def backward(ctx, gy):
        return _bn_backward(ctx, gy, ctx.needs_input_grad[3]) + (None, None, None, None, None, None, None, None)


class _Part:
    """Context of one half of a fused autograd node: what ``_conv_forward`` / ``_bn_forward`` keep between the passes.  The tensors go
    through the node's own ``save_for_backward`` (``_ConvBN``), so nothing here holds a reference to an output of the node."""
    saved_tensors = ()
    needs_input_grad = (True, True, True)

    def save_for_backward(self, *tensors):
        self.saved_tensors = tensors


def _conv_bn_fused_forward(c, b, x, w, bn_w, bn_b, residual, running_mean, running_var, stride, pad, momentum, eps, relu, groups):
    """``_conv_forward`` + ``_bn_forward`` as ONE library call where the convolution runs as F(2x2, 3x3) slabs and the BatchNorm is a
    small-plane one (ResNet layer3 / layer4): fd_conv2d_fwd_bn - the slab reduction happens inside the BatchNorm kernel.  Fills the
    two half-contexts exactly as the separate bodies do; -> (x as float32, conv output, BatchNorm output), or None when the pair does
    not qualify."""
    if groups > 16:
        return None
    xf, wf = f32(x), f32(w)
    plan = _conv_plan(xf, wf, stride, pad, 0, 0, False)
    if not plan.fused_bn_ok(groups):
        return None
    cache_id = getattr(w, "_fd_cache_id", None)
    c.params = (w, None)
    _note_use(w, None)
    b.params = (bn_w, bn_b)
    _note_use(bn_w, bn_b)
    d, nws, nwt = plan.d, plan.fwd_ws, plan.fwd_wt
    y = _empty((d.N, d.Cout, plan.Ho, plan.Wo), xf)
    out = torch.empty_like(y)
    _tally(d, plan.Ho, plan.Wo)
    ws = _empty((nws,), xf)
    wt, ready = _weight_layout(wf, cache_id, "f", nwt, d)
    res = f32(residual) if residual is not None else None
    C = d.Cout
    mean, invstd = _empty((groups * C,), xf), _empty((groups * C,), xf)
    call("fd_conv2d_fwd_bn", plan.dp, ptr(xf), ptr(wf), ptr(y), ptr(wt), ready, ptr(ws), ptr(bn_w), ptr(bn_b), ptr(res), ptr(out),
         ptr(running_mean), ptr(running_var), ptr(mean), ptr(invstd), int(groups), float(eps), float(momentum), int(relu), stream())
    c.save_for_backward(xf, wf, None)
    c.desc, c.has_bias, c.cache_id, c.plan = d, False, cache_id, plan
    b.groups = groups
    b.remask = bool(relu and residual is None and tuning.host.bn_remask)
    b.save_for_backward(y, (bn_b if b.remask else out) if relu else None, bn_w, mean, invstd)
    b.training, b.relu, b.has_res = True, int(relu), residual is not None
    return xf, y, out


class _ConvBN(torch.autograd.Function):
    """Training-mode ``bn(conv(x)) [+ residual] [ReLU]`` of a ResNet block as ONE autograd node (bias-free convolution, zero padding,
    statistics from the convolution's epilogue where its kernel has one): the same C-ABI calls in the same order as ``_Conv2dStats``
    / ``_Conv2dTapStats`` followed by ``_BatchNorm`` - half the ``Function.apply`` calls and backward nodes for the 80 conv + BatchNorm
    pairs of the step (host time only; the launches are unchanged).  ``tap``: also return x for the block's second consumer, whose
    gradient then joins the data gradient in the kernel's epilogue (see ``_Conv2dTap``)."""

    @staticmethod
    def forward(ctx, x, w, bn_w, bn_b, residual, running_mean, running_var, stride, pad, momentum, eps, relu, groups, tap):
        c, b = _Part(), _Part()
        fused = None
        if tuning.host.fused_finish_bn and x.is_cuda:
            fused = _conv_bn_fused_forward(c, b, x, w, bn_w, bn_b, residual, running_mean, running_var, stride, pad, momentum, eps, relu, groups)
        if fused is not None:
            xf, y, out = fused
        else:
            xf, y, part = _conv_forward(c, x, w, None, stride, pad, 0, 0, False, True)
            out = _bn_forward(b, y, bn_w, bn_b, residual, running_mean, running_var, True, momentum, eps, relu, groups, part)
        ctx.n_conv = len(c.saved_tensors)
        ctx.save_for_backward(*(c.saved_tensors + b.saved_tensors))
        c.saved_tensors = b.saved_tensors = ()
        ctx.c, ctx.b = c, b
        ctx.set_materialize_grads(False)
        return (out, xf.view_as(xf)) if tap else out

    @staticmethod
    def backward(ctx, gout, g_tap=None):
        c, b = ctx.c, ctx.b
        saved = ctx.saved_tensors
        c.saved_tensors, b.saved_tensors = saved[:ctx.n_conv], saved[ctx.n_conv:]
        c.needs_input_grad = (ctx.needs_input_grad[0], ctx.needs_input_grad[1], False)
        if gout is None:                              # only the tap was used downstream
            return (f32(g_tap) if g_tap is not None else None,) + (None,) * 13
        gy, gbw, gbb, gres = _bn_backward(b, gout, ctx.needs_input_grad[4])
        gx, gw, _ = _conv_backward(c, gy, g_tap)
        c.saved_tensors = b.saved_tensors = ()
        return (gx, gw, gbw, gbb, gres) + (None,) * 9


_FOLDED = {}               # id(conv weight) -> [stamp, weakref(conv weight), folded weight, folded bias]


def folded_tensors():
    """The folded weight / bias tensors of ``conv_bn_frozen`` (persistent storage for a call recorder)."""
    return [t for e in _FOLDED.values() for t in e[2:4]]


def conv_bn_frozen(x, conv_weight, bn, stride=1, pad=0, relu=False):
    """Eval-mode ``relu?(bn(conv(x)))`` of a FROZEN network (the Refiner's stage-1 ResNets, refiner.py:56-60) with the BatchNorm folded
    into the convolution: w' = w * a[co], b' = bias - running_mean * a, a = weight / sqrt(running_var + eps) (the per-channel constants
    of k_bn_apply_eval), ReLU in the convolution's epilogue - the pass over the activation that applied them is gone.  The folded pair
    is derived once and kept until the weights / statistics change (their version counters, ``invalidate_frozen_layouts``).  The
    result differs from the two-launch form by the rounding of w * a (one ulp per weight)."""
    stamp = (conv_weight._version, _FROZEN_EPOCH[0], conv_weight.data_ptr(), bn.weight._version, bn.bias._version,
             bn.running_mean._version, bn.running_var._version)
    ent = _FOLDED.get(id(conv_weight))
    if ent is None or ent[0] != stamp or ent[1]() is not conv_weight:
        with torch.no_grad():
            a = bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)
            wf = (conv_weight.detach() * a.view(-1, 1, 1, 1)).contiguous()
            bf = (bn.bias.detach() - bn.running_mean * a).contiguous()
        enable_weight_cache([wf], frozen=True)
        key = id(conv_weight)
        ent = _FOLDED[key] = [stamp, weakref.ref(conv_weight, lambda _r, k=key: _FOLDED.pop(k, None)), wf, bf]
    return conv2d(x, ent[2], ent[3], stride, pad, "zero", "relu" if relu else "none")


def conv_bn(x, conv_weight, bn, stride=1, pad=0, residual=None, relu=False, tap=False):
    """``batch_norm(conv2d(x, w), bn, residual, relu)`` in training mode as one autograd node -> out, or (out, x_tap) with ``tap``."""
    groups = _BN_GROUPS[0]
    if bn.num_batches_tracked is not None:
        bump_bn_counter(bn.num_batches_tracked, groups)
    want_tap = bool(tap and x.requires_grad)
    res = _ConvBN.apply(x, conv_weight, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, int(stride), int(pad),
                        bn.momentum, bn.eps, bool(relu), groups, want_tap)
    if tap:
        return res if want_tap else (res, x)
    return res


_BN_GROUPS = [1]
_BN_COUNTERS = [None]


def bump_bn_counter(counter, groups):
    """``num_batches_tracked += groups`` - deferred to one multi-tensor launch inside ``defer_bn_counters``; reported to an active call
    recorder (replay.py replays the bump with the network's recorded calls)."""
    rec = _lib.RECORDER[0]
    if rec is not None:
        rec.side("bn_counter", (counter, groups))
    if _BN_COUNTERS[0] is not None:
        _BN_COUNTERS[0].append((counter, groups))
    else:
        counter.add_(groups)


class defer_bn_counters:
    """Collect the ``num_batches_tracked += groups`` updates of every BatchNorm called inside the block and apply them
    with ONE multi-tensor launch on exit (80 one-element ATen kernels per optimiser step otherwise)."""

    def __enter__(self):
        self.prev = _BN_COUNTERS[0]
        _BN_COUNTERS[0] = []
        return self

    def __exit__(self, *exc):
        pending, _BN_COUNTERS[0] = _BN_COUNTERS[0], self.prev
        if pending and exc[0] is None:
            torch._foreach_add_([t for t, _ in pending], [int(g) for _, g in pending])
        return False



class bn_groups:
    """``with bn_groups(G):`` every training-mode BatchNorm inside treats its batch as G consecutive sub-batches that are
    normalised (and tracked in the running statistics) independently, i.e. exactly like G separate forward passes."""

    def __init__(self, groups):
        self.groups = int(groups)

    def __enter__(self):
        self.prev = _BN_GROUPS[0]
        _BN_GROUPS[0] = self.groups

    def __exit__(self, *a):
        _BN_GROUPS[0] = self.prev


def batch_norm(x, bn, residual=None, relu=False, conv_stats=None):
    """nn.BatchNorm2d semantics (batch statistics + running-stat update in training mode) fused with the
    optional residual add and ReLU.  ``bn`` is an ``nn.BatchNorm2d`` used as the parameter/buffer holder.
    ``conv_stats``: the partial sums ``conv2d_stats`` returned for ``x`` (training mode only)."""
    training = bn.training
    groups = _BN_GROUPS[0] if training else 1
    if training and bn.num_batches_tracked is not None:
        bump_bn_counter(bn.num_batches_tracked, groups)                    # one multi-tensor add per step (trainer)
    return _BatchNorm.apply(x, bn.weight, bn.bias, residual, bn.running_mean, bn.running_var, training, bn.momentum,
                            bn.eps, relu, groups, conv_stats if training else None)


class _BNReluPool(torch.autograd.Function):
    """Training-mode ``maxpool3x3s2(relu(bn(x)))`` of the ResNet stem as one node over fd_bn_relu_maxpool_fwd / _bwd (csrc/norm.hip):
    the full-resolution activation is written only when ``want_feat`` (features[0] has a reader) and never re-read; the backward
    recomputes the ReLU mask / normalised value from x.  Outputs: pooled, or (pooled, feat)."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, groups, want_feat):
        ctx.params = (weight, bias)
        _note_use(weight, bias)
        x = f32(x)
        _need_cuda(x)
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        pooled = _empty((N, C, Ho, Wo), x)
        idx = _empty((N, C, Ho, Wo), x, torch.uint8)
        feat = torch.empty_like(x) if want_feat else None
        mean, invstd = _empty((groups * C,), x), _empty((groups * C,), x)
        ws = _empty((query("fd_bn_ws_floats", N, C, H, W, groups),), x)
        call("fd_bn_relu_maxpool_fwd", ptr(x), ptr(weight), ptr(bias), ptr(feat), ptr(pooled), ptr(idx), ptr(running_mean), ptr(running_var),
             ptr(mean), ptr(invstd), ptr(ws), N, C, H, W, groups, float(eps), float(momentum), stream())
        ctx.save_for_backward(x, weight, bias, mean, invstd, idx)
        ctx.groups = groups
        ctx.set_materialize_grads(False)
        return (pooled, feat) if want_feat else pooled

    @staticmethod
    def backward(ctx, g_pooled, g_feat=None):
        x, weight, bias, mean, invstd, idx = ctx.saved_tensors
        N, C, H, W = x.shape
        if g_pooled is None:                         # only features[0] was used downstream: a zero pooled gradient
            g_pooled = torch.zeros((N, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=x.device, dtype=torch.float32)
        g_pooled = f32(g_pooled)
        g_feat = f32(g_feat) if g_feat is not None else None
        gx = torch.empty_like(x)
        tw, tb = _direct_grad_target(ctx.params[0]), _direct_grad_target(ctx.params[1])
        direct = tw is not None and tb is not None
        gw, gb = (tw, tb) if direct else (_empty((C,), x), _empty((C,), x))
        ws = _empty((query("fd_bn_ws_floats", N, C, H, W, ctx.groups),), x)
        call("fd_bn_relu_maxpool_bwd", ptr(x), ptr(g_pooled), ptr(idx), ptr(g_feat), ptr(weight), ptr(bias), ptr(mean), ptr(invstd), ptr(gx),
             ptr(gw), ptr(gb), ptr(ws), N, C, H, W, ctx.groups, int(direct), stream())
        if direct:
            gw = gb = None
            _grad_ready(ctx.params[0], ctx.params[1])
        return gx, gw, gb, None, None, None, None, None, None


def bn_relu_maxpool(x, bn, want_feature=True):
    """``f0 = relu(bn(x)); pooled = max_pool3x3s2(f0)`` (resnet_encoder.py:95-98) -> (f0 or None, pooled).  Training mode: one fused
    node (``_BNReluPool``); eval mode: the two separate calls."""
    if not bn.training:
        f0 = batch_norm(x, bn, relu=True)
        return f0, max_pool3x3s2(f0)
    groups = _BN_GROUPS[0]
    if bn.num_batches_tracked is not None:
        bump_bn_counter(bn.num_batches_tracked, groups)
    out = _BNReluPool.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, groups, bool(want_feature))
    if not want_feature:
        return None, out
    out[1]._fd_fused_stem = True           # its backward only READS the gradient arriving at features[0] (see mark_single_consumer)
    return out[1], out[0]


def mark_single_consumer(*tensors):
    """The caller guarantees that each of these features[0] tensors of fused stem tails is consumed by ONE node (the depth
    decoder's upsample + concatenation).  ``_UpCat.backward`` then hands both encoders the same gradient tensor instead of cloning
    it: autograd passes a single incoming gradient on without accumulating into it, and ``_BNReluPool.backward`` only reads it."""
    for t in tensors:
        if t is not None and getattr(t, "_fd_fused_stem", False):
            t._fd_grad_readonly = True


class _MaxPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = f32(x)
        _need_cuda(x)
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = _empty((N, C, Ho, Wo), x)
        idx = _empty((N, C, Ho, Wo), x, torch.uint8)
        call("fd_maxpool3x3s2_fwd", ptr(x), ptr(y), ptr(idx), N, C, H, W, stream())
        ctx.save_for_backward(idx)
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        N, C, H, W = ctx.shape
        gx = _empty(ctx.shape, gy)
        call("fd_maxpool3x3s2_bwd", ptr(f32(gy)), ptr(idx), ptr(gx), N, C, H, W, stream())
        return gx


def max_pool3x3s2(x):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)."""
    return _MaxPool.apply(x)


class _UpCat(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, s1, s2, s3, a_act=0, share_skip_grad=False):
        a = f32(a)
        ctx.a_act = int(a_act)
        if ctx.a_act:
            ctx.save_for_backward(a)
        _need_cuda(a)
        N, Ca, h, w = a.shape
        s1 = f32(s1) if s1 is not None else None
        s2 = f32(s2) if s2 is not None else None
        s3 = f32(s3) if s3 is not None else None
        Cs = s1.shape[1] if s1 is not None else 0
        C3 = s3.shape[1] if s3 is not None else 0
        out = _empty((N, Ca + Cs + C3, 2 * h, 2 * w), a)
        call("fd_upcat_fwd", ptr(a), ptr(s1), ptr(s2), ptr(s3), ptr(out), N, Ca, Cs, C3, h, w, stream())
        ctx.dims = (N, Ca, Cs, C3, h, w)
        ctx.has = (s1 is not None, s2 is not None, s3 is not None)
        ctx.share_skip_grad = bool(share_skip_grad)
        return out

    @staticmethod
    def backward(ctx, g):
        N, Ca, Cs, C3, h, w = ctx.dims
        g = f32(g)
        need_s = (ctx.has[0] and ctx.needs_input_grad[1]) or (ctx.has[1] and ctx.needs_input_grad[2])
        ga = _empty((N, Ca, h, w), g) if ctx.needs_input_grad[0] else None
        gs = _empty((N, Cs, 2 * h, 2 * w), g) if need_s else None
        g3 = _empty((N, C3, 2 * h, 2 * w), g) if ctx.has[2] and ctx.needs_input_grad[3] else None
        if ctx.a_act and ga is not None:
            (a_out,) = ctx.saved_tensors
            call("fd_upcat_bwd_act", ptr(g), ptr(a_out), ctx.a_act, ptr(ga), ptr(gs), ptr(g3), N, Ca, Cs, C3, h, w, stream())
        else:
            call("fd_upcat_bwd", ptr(g), ptr(ga), ptr(gs), ptr(g3), N, Ca, Cs, C3, h, w, stream())
        # skip and skip_add come from encoders that run their backward on DIFFERENT streams.  Handing both the same tensor is
        # a race: autograd may accumulate a second incoming gradient into it in place on one stream while the other stream's
        # kernels still read it (seen as run-to-run different encoder gradients under GPU contention).  One owner each.
        # ``share_skip_grad``: both skips are features[0] of fused stem tails (_BNReluPool): each has this node as its ONLY consumer, so
        # autograd hands the tensor on without accumulating into it, and the stem-tail backward kernels only read it - no copy.
        both = ctx.has[0] and ctx.has[1] and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and not ctx.share_skip_grad
        return ga, gs if ctx.has[0] else None, (gs.clone() if both else gs) if ctx.has[1] else None, g3, None, None


def upsample_concat(a, skip=None, skip_add=None, extra=None, a_act="none"):
    """cat([nearest_up2(a), skip (+ skip_add), extra], 1)  — depth_decoder.py:75-83 in one pass.  ``a_act``: ``a`` is the output of
    that activation, produced by a ``conv2d(..., grad_preact=True)`` and consumed here alone: its gradient is returned times act'(a)."""
    share = bool(getattr(skip, "_fd_grad_readonly", False) and getattr(skip_add, "_fd_grad_readonly", False))
    if ACT[a_act] and torch.is_grad_enabled() and a.requires_grad:
        return _UpCat.apply(a, skip, skip_add, extra, ACT[a_act], share)
    return _UpCat.apply(a, skip, skip_add, extra, 0, share)


class _Up2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = f32(x)
        _need_cuda(x)
        N, C, h, w = x.shape
        y = _empty((N, C, 2 * h, 2 * w), x)
        call("fd_upsample2x_fwd", ptr(x), ptr(y), N * C, h, w, stream())
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, g):
        N, C, h, w = ctx.shape
        gx = _empty(ctx.shape, g)
        call("fd_upsample2x_bwd", ptr(f32(g)), ptr(gx), N * C, h, w, stream())
        return gx


def upsample_nearest2x(x):
    return _Up2.apply(x)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = f32(a), f32(b)
        _need_cuda(a, b)
        out = torch.empty_like(a)
        call("fd_axpby", ptr(a), ptr(b), ptr(out), a.numel(), 1.0, 1.0, stream())
        return out

    @staticmethod
    def backward(ctx, g):
        # the two operands belong to encoders on different streams: separate gradient tensors (see _UpCat.backward)
        return g, (g.clone() if ctx.needs_input_grad[0] and ctx.needs_input_grad[1] else g)


def add(a, b):
    """Feature fusion a + b (depth_decoder.py:70, pose_decoder.py:31)."""
    return _Add.apply(a, b)


class _SpatialMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        x = f32(x)
        _need_cuda(x)
        N, C, H, W = x.shape
        out = _empty((N, C), x)
        call("fd_spatial_mean_fwd", ptr(x), ptr(out), N * C, H * W, float(scale), stream())
        ctx.shape, ctx.scale = x.shape, float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.shape
        gx = _empty(ctx.shape, g)
        call("fd_spatial_mean_bwd", ptr(f32(g)), ptr(gx), N * C, H * W, ctx.scale, stream())
        return gx, None


def spatial_mean(x, scale=1.0):
    """scale * x.mean(3).mean(2)  (pose_decoder.py:44-46)."""
    return _SpatialMean.apply(x, scale)


def resize_linear_cv(x, size):
    """``cv2.resize(img, (size[1], size[0]))`` (INTER_LINEAR, float32) for every plane of ``x`` [..., H, W] -> [..., size[0], size[1]]
    (evaluate_depth.py:349; OpenCV's coefficient rule, see fd_resize_linear_cv)."""
    x = f32(x.detach())
    _need_cuda(x)
    H, W = x.shape[-2:]
    planes = x.numel() // (H * W)
    y = _empty(tuple(x.shape[:-2]) + (int(size[0]), int(size[1])), x)
    call("fd_resize_linear_cv", ptr(x), ptr(y), planes, H, W, int(size[0]), int(size[1]), stream())
    return y


def masked_median(x, gate, scale=1.0, window=None):
    """``torch.median(x[mask] * scale)`` with ``mask = gate > 0`` inside ``window`` = (y0, y1, x0, x1) of every plane (default: the
    whole plane), over the whole batch - refiner.py:327-331.  A 0-dim device tensor; no host round trip, no sort."""
    x, gate = f32(x.detach()), f32(gate.detach())
    _need_cuda(x, gate)
    if x.shape != gate.shape:
        raise RuntimeError("masked_median: x %s and gate %s differ in shape" % (tuple(x.shape), tuple(gate.shape)))
    H, W = x.shape[-2:]
    B = x.numel() // (H * W)
    y0, y1, x0, x1 = window if window is not None else (0, H, 0, W)
    out = _empty((2,), x)
    ws = torch.empty((query("fd_masked_median_ws_bytes", B, H, W),), device=x.device, dtype=torch.uint8)
    call("fd_masked_median", ptr(x), ptr(gate), float(scale), B, H, W, int(y0), int(y1), int(x0), int(x1), ptr(out), ws.data_ptr(), stream())
    return out[0]


def refine_inputs(disps, beam, two_cha, inv_Ks, height, width, min_depth, max_depth, catxy=True, pool_disp0=True,
                  crop=(78, 190, 23, 617), return_stats=False):
    """refiner.py:316-348 for all scales in four launches (csrc/refine.hip): -> list of [B, 1 + 3*catxy + 2, Hs, Ws] tensors, the
    ``depth_maps`` inputs of the refine decoder (scaled disparity | Cat_xy of the pooled depth | pooled 2-channel LiDAR map).
    ``disps``: ``outputs[("disp", s)]`` per scale (only ``disps[0]`` is read when ``pool_disp0`` = --refine_a0 true);
    ``inv_Ks``: ``inputs[("inv_K", s)]`` per scale.  No gradient (the reference computes all of it under no_grad)."""
    S = len(disps)
    d0 = f32(disps[0].detach())
    beam, two_cha = f32(beam.detach()), f32(two_cha.detach())
    _need_cuda(d0, beam, two_cha)
    B = d0.shape[0]
    cfg = _lib.RefineCfg()
    cfg.B, cfg.H, cfg.W, cfg.n_scales = B, int(height), int(width), S
    ds, ks = [], []
    for s in range(S):
        cfg.Hs[s], cfg.Ws[s] = (int(disps[s].shape[2]), int(disps[s].shape[3]))
        ds.append(d0 if s == 0 else (None if pool_disp0 else f32(disps[s].detach())))
        ks.append(f32(inv_Ks[s].detach()) if catxy else None)
    # the reference writes `crop_mask[:, :, 78:190, 23:617] = 1` (refiner.py:329): a slice, i.e. silently clamped to the plane - at 96x320 or
    # 128x416 the window shrinks, below 79 rows it is empty (medians NaN, as torch.median of an empty selection); ADVICE round 5
    y0, y1, x0, x1 = (int(v) for v in crop)
    y1, x1 = min(y1, int(height)), min(x1, int(width))
    y0, x0 = min(y0, y1), min(x0, x1)
    cfg.crop_y0, cfg.crop_y1, cfg.crop_x0, cfg.crop_x1 = y0, y1, x0, x1
    cfg.min_depth, cfg.max_depth, cfg.catxy, cfg.pool_disp0 = float(min_depth), float(max_depth), int(bool(catxy)), int(bool(pool_disp0))
    C = 1 + (3 if catxy else 0) + 2
    outs = [_empty((B, C, cfg.Hs[s], cfg.Ws[s]), d0) for s in range(S)]
    ws = torch.empty((query("fd_refine_inputs_ws_bytes", ctypes.byref(cfg)),), device=d0.device, dtype=torch.uint8)
    stats = _empty((S, 4), d0) if return_stats else None
    arr = lambda ts: (ctypes.c_void_p * 4)(*[ptr(t) for t in ts] + [None] * (4 - len(ts)))
    call("fd_refine_inputs", ctypes.byref(cfg), arr(ds), ptr(beam), ptr(two_cha), arr(ks), arr(outs), ptr(stats), ws.data_ptr(), stream())
    return (outs, stats) if return_stats else outs


def depth_errors(gt, pred):
    """layers.py:284-302 on matched 1-D tensors -> 7 scalars (abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3)."""
    gt, pred = f32(gt.detach()).reshape(-1), f32(pred.detach()).reshape(-1)
    _need_cuda(gt, pred)
    out, ws = _empty((7,), gt), _empty((7 * 256,), gt)
    call("fd_depth_errors", ptr(gt), ptr(pred), gt.numel(), ptr(out), ptr(ws), stream())
    return tuple(out[i] for i in range(7))


def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """One torch.optim.Adam update of a flat fp32 tensor, in place."""
    bc1, bc2 = 1.0 - betas[0] ** step, 1.0 - betas[1] ** step
    _sync_late_layouts()
    bump_weights_epoch()
    call("fd_adam_step", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), float(lr), betas[0],
         betas[1], float(eps), bc1, bc2, float(grad_scale), stream())
    refresh_weight_layouts()


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, state, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
    """Adam update whose step counter / lr live in ``state`` (device, [step, lr]) — hipGraph-replay safe."""
    _sync_late_layouts()
    bump_weights_epoch()
    call("fd_adam_step_dev", ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), ptr(state), betas[0],
         betas[1], float(eps), float(grad_scale), stream())
    refresh_weight_layouts()
