/* libfdhip — C ABI of the MI355X-native (gfx950) FusionDepth training hot path.
 *
 * The reference (AutoAILab/FusionDepth) has no FFI layer: its boundary is the Python module API in
 * layers.py / networks/*.py / trainer.py, whose arithmetic is implicit ATen/cuDNN calls.  Each entry
 * point below replaces one such call site (cited as reference file:line).  `fusiondepth_amd` binds
 * these with ctypes (fusiondepth_amd/_lib.py); INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every tensor is float32, contiguous, NCHW unless stated; pointers are DEVICE pointers owned by
 *     the caller (PyTorch caching allocator); the library never allocates device memory and keeps
 *     no state besides the last-error string.  Workspaces are passed in with their size in floats.
 *   - `stream` is a hipStream_t; every call is asynchronous and stream-ordered, no implicit sync.
 *   - return 0 on success; a hipError_t (>0) for launch failures; -1 for bad arguments.  Never
 *     throws/aborts.  fd_last_error() describes the last failure on the calling thread.
 *   - reductions are deterministic (fixed trees, no float atomics).
 */
#ifndef FDHIP_H
#define FDHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: fd_tuning / fd_set_tuning added (the library no longer reads the process environment); fd_conv2d_fwd_stats and
 *    fd_bn_train_fwd_parts added, the fd_conv2d_*_pair entry points removed, fd_bn_ws_floats grew by one shift value per
 *    (group, channel) - a client that sized the BatchNorm workspace itself must re-query it.
 * 3: additions only (round 5): fd_masked_median, fd_refine_inputs (+ fd_refine_cfg), fd_resize_linear_cv, fd_bn_relu_maxpool_fwd / _bwd,
 *    fd_bn_train_bwd_remask, fd_stack_normalize, fd_conv2d_fwd_bn(_ok), fd_pose_head_fwd / _bwd; fd_tuning grew at its end
 *    (wino_min_cout, wino_wgrad_min_cout, wino_wgrad_xcd_few, wino_fwd_halfm, wino_wgrad_halfm, grp_tile64_below).  Nothing removed, no signature changed.
 * 4: additions only (round 6): fd_replay (+ fd_call_rec, fd_replay_function_count / _name / _signature); fd_tuning grew at its end (limb_1x1,
 *    limb_depth, limb_target, limb_split_max_out, limb_wgrad_target, limb_conv, wino_wgrad_limb, wino_fwd_limb); fd_relayout_job.mode 7 / 8 (1x1 weights pre-split into bf16
 *    limbs) and 9 / 10 (the same for a tap subset of a larger kernel); fd_refine_cfg accepts an empty crop window.  Nothing removed, no
 *    signature changed. */
#define FD_ABI_VERSION 4

int fd_abi_version(void);
const char* fd_supported_arch(void); /* "gfx950" */
const char* fd_last_error(void);

/* ------------------------------------------------------------------ tuning ----------------------
 * Kernel-selection thresholds, process-wide.  The library NEVER reads the process environment: its behaviour - which kernel
 * family a convolution is routed to, and therefore what fd_*_wt_floats / fd_*_ws_floats return - is a function of the arguments
 * and of this struct alone.  The defaults are the measured best on MI355X and are in force from load on; the other values exist
 * for A/B timing and for tests that pin a kernel family.  fd_set_tuning copies the struct (fields beyond `size` bytes keep their
 * defaults); it must not race with calls in flight on other threads, and size queries made before the change do not apply after
 * it (a caller that changes the tuning re-queries its workspaces; fd_tuning_generation() counts the changes). */
typedef struct fd_tuning {
    int size;                     /* sizeof(fd_tuning) as the caller compiled it */
    int wino_fwd;                 /* 1   3x3 stride-1 forward / data gradient on the Winograd kernels (0: direct implicit GEMM) */
    int wino_wgrad;               /* 1   their weight gradient on the transposed Winograd kernel */
    int wino_fwd_2d_min;          /* 65536   F(2x2,3x3) forward / data gradient from Cin*Cout on (0: F(2,3) along x everywhere) */
    int wino_fwd_2dp_min_wgs;     /* 160 below that Cin*Cout: F(2x2,3x3) with all 16 components in ONE workgroup (k_conv_wino2p, no slabs) where
                                         the launch has at least this many 64x64 output tiles (0: never - F(2,3) along x as in rounds 1-3) */
    int wino_fwd_2dp_dma;         /* 1   ... with its activations on the direct-to-LDS path where W % 4 == 0 (0: register-staged loader) */
    int wino_fwd_2dp_deep;        /* 0   1: the one-workgroup kernel also above wino_fwd_2d_min when the launch has enough tiles (A/B) */
    int wino_wgrad_2d;            /* 2   weight gradient as transposed F(2x2,3x3) where the height is even (1: and Cin % 32 == 0; 0: never) */
    int wino_target;              /* 384 workgroups a Winograd forward launch is split-K'd up to */
    int wino_wgrad_target;        /* 256 ... a Winograd weight-gradient launch is pixel-sliced up to */
    int conv_target;              /* 768 ... a direct forward / data-gradient launch */
    int wgrad_target;             /* 768 ... a direct weight-gradient launch */
    int conv_c1;                  /* 1   Cout == 1 layers (dispconv) as stencils (conv_c1.hip) */
    int conv_n16_min_pixels;      /* 16384   16 / 32-channel 3x3 blocks on k_conv3x3_n16 from this plane size on (< 0: never) */
    int reflect_ring;             /* 1   reflect-padded data gradients as interior + ring from 16384 pixels on (n > 1: from n pixels
                                         on; 0: always through the padded-grid tensor + fold pass) */
    int reflect_wino;             /* 1   the interior of such a gradient on the Winograd kernels (>= 64 input channels) */
    int reflect_wino_min_pixels;  /* 1   ... on planes of at least this many pixels */
    int reflect_wino_padded_max;  /* 4096   below this plane size: ONE Winograd convolution over the zero-bordered dY + fold (0: never) */
    int force_cfg, force_splits;  /* -1, 1   force the direct kernel's tile configuration (0..2) and split-K (sweeps) */
    int stem7;                    /* 1   7x7 stride-2 stems (Cin 2..6) on the dedicated patch kernels (conv_stem.hip) */
    int log;                      /* 0   1: one stderr line per convolution call with the kernel family it was routed to */
    int wino_fwd_2d_m128;         /* 1   the slab variant with 128 output channels per workgroup (k_conv_wino2d_m128) wherever it can run (Cout % 128 == 0, W % 4 == 0); 2: only where its launch fills the chip better; 0: never */
    int wino_min_cout;            /* 32  fewest output channels of a forward / data-gradient launch on the Winograd kernels (their tile is 64 channels
                                         tall: below that, rows of the tile are idle; 32 = the depth decoder's upconv(1, *) as well; rounds 1-4: 64) */
    int wino_wgrad_min_cout;      /* 32  ... of a weight-gradient launch */
    int wino_wgrad_xcd_few;       /* 1   Winograd weight gradients with 2 or 4 pixel slices on the XCD-aware grid as well (each slice owns 4 / 2 XCDs) */
    int wino_fwd_halfm;           /* 1   k_conv_wino2p_dma with <= 32 output channels likewise (forward of the decoder's upconv(1, *)) */
    int wino_wgrad_halfm;         /* 1   Winograd weight gradients with <= 32 output channels: the tile's idle wave pair takes half of every chunk's K-steps */
    int grp_tile64_below;         /* 0   grouped launches (the four parity classes of a stride-2 data gradient) with fewer than this many 64x128
                                         workgroups run on 64x64 tiles (twice the workgroups, half the work of the longest one); 0: never */
    int limb_1x1;                 /* 1   1x1 stride-1 convolutions with >= 64 channels on both sides (ResNet-50 bottlenecks) as split-precision GEMMs on the
                                         bf16 MFMA: every fp32 operand = 3 bf16 limbs, six limb products, fp32 accumulation - fp32 accuracy
                                         (csrc/conv_limb.hip); 0: the f32-MFMA direct kernels */
    int limb_depth;               /* 2   K-chunks (16 channels each) of operands a thread of the limb GEMMs keeps in flight (2 or 4) */
    int limb_target;              /* 256 workgroups a limb forward / data-gradient launch is split-K'd up to ... */
    int limb_split_max_out;       /* 4194304   ... when its output has at most this many floats (a split costs a slab round trip of the output) */
    int limb_wgrad_target;        /* 256 workgroups a limb weight-gradient launch is pixel-sliced up to (x2 for its 4-wave tiles) */
    int limb_conv;                /* 1   stride-2 convolutions with >= 64 channels on both sides (ResNet layerN.0.conv1 / downsample: no Winograd form), forward
                                         and data gradient, as split-precision implicit GEMMs (k_conv_limb); 0: the f32-MFMA direct kernels */
    int wino_wgrad_limb;          /* 2   the 2-D Winograd weight gradient with >= 64 output channels and W % 8 == 0 with a split-precision matrix loop
                                         (k_wgrad_wino_limb: transforms + limb split in the loader, 768 instead of 2 048 matrix cycles per chunk): 1 the
                                         zero-padded layers (ResNet trunk), 2 the reflect-padded decoder blocks as well, 0 the f32 kernel everywhere */
    int wino_fwd_limb;            /* 0   1: the F(2x2, 3x3) slab kernel of the deep layers (forward / data gradient, >= 64 channels on both sides, W % 4 == 0)
                                         with pre-split weights and a split-precision matrix loop (k_conv_wino2d_limb) instead of k_conv_wino2d(_m128).
                                         Per-kernel error vs float64 equal to the f32 kernels' and +0.2 ... +0.8 % in the step, but OFF: with it one bound
                                         of the full-size backward test at 1024x320 is exceeded (depth-decoder gradient norm 1.81e-3 against 1.65e-3) */
} fd_tuning;
void fd_tuning_defaults(fd_tuning* t);
int fd_set_tuning(const fd_tuning* t);
void fd_get_tuning(fd_tuning* t);
long fd_tuning_generation(void);

/* ------------------------------------------------------------------ geometry (layers.py) ------- */

/* layers.py:11-20 disp_to_depth.  depth and/or scaled may be NULL. */
int fd_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, long n, double min_depth, double max_depth,
                         void* stream);
/* d_disp = g_scaled*(max_disp-min_disp) - g_depth*(max_disp-min_disp)*depth^2 ; either g may be NULL */
int fd_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* d_disp, long n,
                         double min_depth, double max_depth, void* stream);

/* layers.py:23-97 transformation_from_parameters (+rot_from_axisangle, get_translation_matrix).
 * axisangle, translation: [B,3]; T: [B,4,4].  invert: M = R^T * T(-t), else T(t) * R. */
int fd_pose_matrix_fwd(const float* axisangle, const float* translation, float* T, int B, int invert, void* stream);
int fd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* gT, float* g_axisangle,
                       float* g_translation, int B, int invert, void* stream);

/* trainer.py:338-360 for the STACKED pose network (all frame pairs x accumulated micro-batches in one batch): pose [G*nf*Bq][ld] = the
 * pose decoder's output (networks/pose_decoder.py:47-51: 0.01 * mean, ld = 6 * predictions; prediction 0 = columns 0..5 is the one
 * trainer.py:350-352 uses), rows ordered (micro-batch g, frame pair k, sample s) -> per frame pair k: T[k] [G*Bq][4][4] (bit k of
 * invert_mask: trainer.py:352 invert=(f_i < 0)) and, if given, axisangle[k] / translation[k] [G*Bq][ld/6][3] (the outputs dictionary's
 * entries).  T / axisangle / translation / gT: HOST arrays of nf (<= 4) device pointers.  Replaces, per frame pair, the slicing +
 * concatenation + fd_pose_matrix_fwd / _bwd launches (and autograd's ~20 element-wise launches per pair behind them).
 * bwd: g_pose [G*nf*Bq][ld], fully written (unused predictions: 0); gT[k] == NULL: that matrix received no gradient. */
int fd_pose_head_fwd(const float* pose, float* const* T, float* const* axisangle, float* const* translation, int G, int nf, int Bq,
                     int ld, unsigned invert_mask, void* stream);
int fd_pose_head_bwd(const float* pose, const float* const* gT, float* g_pose, int G, int nf, int Bq, int ld, unsigned invert_mask,
                     void* stream);

/* layers.py:217 `P = matmul(K, T)[:, :3, :]`.  K,T: [B,4,4]; P: [B,3,4] written at P + b*p_batch_stride. */
int fd_proj_matrix_fwd(const float* K, const float* T, float* P, long p_batch_stride, int B, void* stream);
/* gT[B,4,4] = K[:3,:]^T * gP */
int fd_proj_matrix_bwd(const float* K, const float* gP, long p_batch_stride, float* gT, int B, void* stream);

/* layers.py:157-162 BackprojectDepth.forward: depth[B,1,H,W], inv_K[B,4,4] -> points [B,4,H*W]. */
int fd_backproject_fwd(const float* depth, const float* inv_K, float* points, int B, int H, int W, void* stream);
int fd_backproject_bwd(const float* g_points, const float* inv_K, float* g_depth, int B, int H, int W, void* stream);

/* layers.py:215-226 Project3D.forward: points[B,4,H*W], K, T -> grid [B,H,W,2]. */
int fd_project3d_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W, float eps,
                     void* stream);
/* g_points [B,4,HW] (row 3 gets its true gradient too); gP partial sums need `ws` of
 * fd_project3d_bwd_ws_floats(B,H,W) floats; gT [B,4,4]. */
long fd_project3d_bwd_ws_floats(int B, int H, int W);
int fd_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid, float* g_points,
                     float* gT, float* ws, int B, int H, int W, float eps, void* stream);

/* layers.py:187-201 Cat_xy.forward (refiner.py input channels). out [B,3,H,W]. */
int fd_cat_xy_fwd(const float* depth, const float* inv_K, float* out, int B, int H, int W, void* stream);

/* F.interpolate(x,[H,W],mode="bilinear",align_corners=False) (trainer.py:434-435,:579), C channels. */
int fd_bilinear_up_fwd(const float* x, float* y, int BC, int Hin, int Win, int Hout, int Wout, void* stream);
int fd_bilinear_up_bwd(const float* gy, float* gx, int BC, int Hin, int Win, int Hout, int Wout, void* stream);

/* ------------------------------------------------------------------ photometric loss ----------- */

/* layers.py:267-281 SSIM.forward -> out [B,C,H,W] = clamp((1-SSIM)/2,0,1). */
int fd_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, void* stream);
/* gradients of sum(out*g) w.r.t. x and y (either may be NULL). */
int fd_ssim_bwd(const float* x, const float* y, const float* g, float* gx, float* gy, int B, int C, int H, int W,
                void* stream);

/* trainer.py:476-488 compute_reprojection_loss (3-channel images): out[b] at out + b*out_batch_stride,
 * = 0.85*mean_c SSIM + 0.15*mean_c|t-p|  (use_ssim) or mean_c|t-p|. */
int fd_reproj_loss_map(const float* pred, const float* target, float* out, long out_batch_stride, int B, int H, int W,
                       int use_ssim, void* stream);

/* Fused per-scale photometric + LiDAR loss, forward.  Replaces, for one scale, trainer.py:434-470
 * (bilinear upsample of disp, disp_to_depth, BackprojectDepth, Project3D, F.grid_sample(border)) and
 * trainer.py:509-567,577-589 (SSIM+L1 reprojection losses, identity losses + noise, per-pixel min,
 * mean; masked scale-invariant log loss vs the 4-beam LiDAR).
 *
 *   disp      [B,1,Hs,Ws]   sigmoid output of the decoder at this scale
 *   inv_K     [B,4,4]       inverse intrinsics at the sampling resolution (H,W)
 *   P         [B,NF,3,4]    (K @ cam_T_cam_f)[:3] per source frame (fd_proj_matrix_fwd)
 *   src       NF pointers   source colour images [B,3,H,W]
 *   target    [B,3,H,W]
 *   ident     [B,NI,H,W]    identity reprojection losses (NI = NF, or 1 if avg_reprojection) or NULL
 *   noise     [B,NI,H,W]    tie-break noise (scaled by 1e-5 inside) or NULL
 *   beam      [B,1,H,W]     4-beam LiDAR depth / 100, or NULL to skip the SI loss
 *   sel       [B,H,W] u8    out: argmin index into cat(ident, reproj) (trainer.py:561)
 *   depth_out/sample_out/color_out   optional materialised ("depth",0,s) [B,1,H,W],
 *                           ("sample",f,s) [NF][B,H,W,2], ("color",f,s) [NF][B,3,H,W]  (NULL to skip)
 *   ws        workspace of fd_photo_ws_floats(B,H,W) floats (per-block partial sums)
 *   out       [FD_PHOTO_OUT_FLOATS] floats: 0 to_optimise.mean() over the whole batch, 4 si_loss averaged over the
 *             `groups` sub-batches, 8+4g.. per group g: n_valid, mean(d), mean(d^2)-si_var*mean(d)^2, si_loss_g
 *             (1..3 repeat group 0; the rest is scratch)
 *   cfg.groups  G >= 1: the batch is G stacked micro-batches (trainer.py:237-248 accumulates their losses); the SI-log
 *             loss, which is not linear in the batch, is evaluated per micro-batch and averaged.
 */
#define FD_PHOTO_OUT_FLOATS 96
typedef struct {
    double min_depth, max_depth; /* opt.min_depth / opt.max_depth (doubles: 1/0.1 must be exactly 10) */
    int B, H, W, Hs, Ws, NF;
    int use_ssim;        /* !opt.no_ssim */
    int avg_reprojection;
    float si_depth_scale;   /* 26.0  (trainer.py:583) */
    float si_beam_scale;    /* 100.0 (trainer.py:581) */
    float si_threshold;     /* opt.gdc_loss_threshold */
    float si_var;           /* opt.si_var */
    float eps;              /* Project3D eps 1e-7 */
    int groups;             /* number of stacked micro-batches (>= 1) */
    float si_lo;            /* lower bound of the SI-loss mask: target > si_lo && pred > si_lo  (1.0 in trainer.py:584-586,
                               1e-3 in refiner.py:561-562) */
    int si_mode;            /* LiDAR term: 0 = scale-invariant log loss * 0.1 (trainer.py:577-589, completor.py:699-716),
                               1 = masked L1 * 0.001 without the |pred - beam| gate (completor.py:718-723, --completion_l1loss);
                               out[8+4g+1] then holds mean |pred - beam| */
} fd_photo_cfg;

long fd_photo_ws_floats(int B, int H, int W);
int fd_photo_fwd(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                 const float* const* src, const float* target, const float* ident, const float* noise,
                 const float* beam, uint8_t* sel, float* depth_out, float* sample_out, float* color_out, float* ws,
                 float* out, void* stream);
/* Backward of the above.  g [2] device floats: dL/d(out[0]), dL/d(out[4]).  stats = `out` of the
 * forward; has_ident = whether the forward was given `ident` (needed to decode `sel`).
 * d_disp [B,1,Hs,Ws]; gP [B,NF,3,4]; ws: fd_photo_bwd_ws_floats(B,H,W) floats. */
long fd_photo_bwd_ws_floats(int B, int H, int W);
int fd_photo_bwd(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                 const float* const* src, const float* target, const float* beam, const uint8_t* sel, int has_ident,
                 const float* stats, const float* g, float* d_disp, float* gP, float* ws, void* stream);
/* The same pair with the remaining flag variants of trainer.py:425-567: up to THREE source frames (--use_stereo adds the
 * stereo partner "s" to frames -1 / +1, trainer.py:436-439) and the predictive-mask baseline (--predictive_mask,
 * trainer.py:117-127, 530-541).
 *   mask        [B,NF,H,W] or NULL: the reprojection loss of frame f is multiplied by mask[:, f] before the minimum /
 *               average (the mask replaces automasking: `ident` must be NULL when it is given)
 *   reproj_out  [B,NF,H,W] or NULL: the UNmasked reprojection losses; d out[0] / d mask[:, f] = reproj_out[:, f] / (B H W)
 *               where frame f was selected (`sel`), or / NF everywhere with avg_reprojection - the caller forms it.
 * fd_photo_fwd / fd_photo_bwd are these with mask = reproj_out = NULL. */
int fd_photo_fwd_ex(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                    const float* const* src, const float* target, const float* ident, const float* noise,
                    const float* beam, const float* mask, uint8_t* sel, float* depth_out, float* sample_out,
                    float* color_out, float* reproj_out, float* ws, float* out, void* stream);
int fd_photo_bwd_ex(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                    const float* const* src, const float* target, const float* beam, const float* mask, const uint8_t* sel,
                    int has_ident, const float* stats, const float* g, float* d_disp, float* gP, float* ws, void* stream);

/* ---- all pyramid scales in one launch, value + unit-cotangent gradient in the same pass ------------------------------
 * Replaces the per-scale loop of trainer.py:425-474 (generate_images_pred) + trainer.py:509-567, 577-589 (compute_losses)
 * for the default configuration: two source frames, SSIM + L1, per-frame minimum (automasking on or off), SI-log or masked
 * L1 LiDAR term on any subset of the scales.  The flag variants (--no_ssim, --avg_reprojection, one source frame, the
 * materialised ("depth"|"sample"|"color", f, s) outputs) stay on fd_photo_fwd / fd_photo_bwd.
 *   cfg.base      as for fd_photo_fwd (Hs / Ws ignored: per-scale sizes are cfg.Hs[] / cfg.Ws[]); NF must be 2
 *   disp[s]       [B,1,Hs[s],Ws[s]]      noise[s]  [B,2,H,W] or NULL (the array itself may be NULL)
 *   ident         [B,2,H,W] or NULL      beam      [B,1,H,W] or NULL; cfg.beam_mask bit s = scale s carries the LiDAR term
 *   sel           [S,B,H,W] u8 out       d1        [S,B,H,W] out: d out[s][0] / d disp_up, or NULL to skip every gradient
 *   ws            fd_photo_ms_ws_floats(cfg) floats; must stay untouched until fd_photo_ms_bwd has run
 *   out           [S][FD_PHOTO_OUT_FLOATS], each laid out like fd_photo_fwd's `out`
 * fd_photo_ms_bwd: g_photo[s] / g_si[s] = device pointers to dL/d out[s][0] / dL/d out[s][4] (NULL = 0); d1 is read only;
 * d_disp[s] [B,1,Hs[s],Ws[s]] out (H / Hs[s] == W / Ws[s] must be an integer <= 16);
 * gP [B,2,3,4] out = sum over scales of g_photo[s] * d out[s][0] / d P. */
typedef struct {
    fd_photo_cfg base;
    int n_scales;            /* 1..4 */
    int Hs[4], Ws[4];
    unsigned beam_mask;
    int rows_per_strip;      /* image rows a wave streams through; 0 = default */
} fd_photo_ms_cfg;
long fd_photo_ms_ws_floats(const fd_photo_ms_cfg* cfg);
int fd_photo_ms_fwd(const fd_photo_ms_cfg* cfg, const float* const* disp, const float* inv_K, const float* P,
                    const float* const* src, const float* target, const float* ident, const float* const* noise,
                    const float* beam, uint8_t* sel, float* d1, float* ws, float* out, void* stream);
int fd_photo_ms_bwd(const fd_photo_ms_cfg* cfg, const float* const* disp, const float* beam, const float* stats,
                    const float* const* g_photo, const float* const* g_si, float* d1, const float* ws,
                    float* const* d_disp, float* gP, void* stream);

/* layers.py:235-248 get_smooth_loss on the mean-normalised disparity (trainer.py:569-571):
 * out[0] = get_smooth_loss(disp/(mean_hw(disp)+1e-7), img).  ws: fd_smooth_ws_floats(B,H,W). */
long fd_smooth_ws_floats(int B, int H, int W);
int fd_smooth_fwd(const float* disp, const float* img, float* out, float* ws, int B, int H, int W, int normalize,
                  void* stream);
/* d_disp [B,1,H,W] = g[0] * d out[0] / d disp */
int fd_smooth_bwd(const float* disp, const float* img, const float* g, float* d_disp, float* ws, int B, int H, int W,
                  int normalize, void* stream);

/* ------------------------------------------------------------------ convolution stack ---------- */

/* One 2-D convolution, NCHW fp32, square kernel 1/3/5/7, stride 1|2, symmetric padding.
 * Replaces nn.Conv2d / layers.Conv3x3 / layers.ConvBlock call sites:
 *   networks/resnet_encoder.py:95,98-101 (torchvision ResNet convs, zero pad, no bias),
 *   layers.py:115-130 (ReflectionPad2d(1)+Conv2d 3x3 + bias), layers.py:100-112 (+ELU),
 *   networks/depth_decoder.py:92 (dispconv + sigmoid), networks/pose_decoder.py:35-42 (+ReLU). */
typedef struct {
    int N, Cin, H, W;       /* input  [N,Cin,H,W]                    */
    int Cout, KH, KW;       /* weight [Cout,Cin,KH,KW] (OIHW)        */
    int stride, pad;
    int pad_mode;           /* 0 zero, 1 reflect (stride 1, pad 1)   */
    int act;                /* epilogue: 0 none, 1 ReLU, 2 ELU, 3 sigmoid, 4 tanh (applied after bias) */
    int in_norm;            /* 1: taps read (x-0.45)/0.225 (resnet_encoder.py:94), padding stays 0 */
} fd_conv_desc;

/* y [N,Cout,Ho,Wo] = act(conv(x, w) + bias);  bias may be NULL.
 *   wt  caller-owned buffer of fd_conv2d_fwd_wt_floats(d) floats holding the kernel's weight layout ([Cout][tap][Cin]);
 *       it is (re)written from `w` unless wt_ready != 0, so a caller may keep it across calls while `w` is unchanged
 *       (the trainer re-derives it once per optimiser step).  May be NULL when the size query returns 0.
 *   ws  scratch of fd_conv2d_fwd_ws_floats(d) floats (split-K slabs; may be 0 -> NULL). */
long fd_conv2d_fwd_wt_floats(const fd_conv_desc* d);
long fd_conv2d_fwd_ws_floats(const fd_conv_desc* d);
int fd_conv2d_fwd(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                  int wt_ready, float* ws, void* stream);
/* BatchNorm statistics from the convolution's own epilogue.  Every ResNet convolution is followed by a training-mode BatchNorm
 * (networks/resnet_encoder.py:95-101 -> torchvision BasicBlock / Bottleneck), whose first pass - per-channel sum and sum of
 * squares of the convolution output - the convolution kernel can produce while the output tile is still in registers:
 *   fd_conv2d_fwd_stat_slots  S = partial-sum slots per (image, output channel) the kernel chosen for `d` writes, or 0 when that
 *                             kernel has no statistics epilogue (split-K launches, tiles that straddle images, fused activations):
 *                             the caller then uses fd_bn_train_fwd, which makes its own statistics pass.
 *   fd_conv2d_fwd_stats       fd_conv2d_fwd that also fills part [N][Cout][S][2] = (sum, M2) of y per slot of H*W/S pixels,
 *                             M2 = sum of squared deviations from the slot's own mean (merged without cancellation).
 *   fd_bn_train_fwd_parts     fd_bn_train_fwd (same semantics: per-group statistics, running statistics updated in group order)
 *                             with the statistics taken from such partial sums - one launch, one pass over x less. */
long fd_conv2d_fwd_stat_slots(const fd_conv_desc* d);
int fd_conv2d_fwd_stats(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                        int wt_ready, float* ws, float* stat_part, void* stream);
int fd_bn_train_fwd_parts(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                          float* running_mean, float* running_var, float* save_mean, float* save_invstd, const float* conv_part,
                          int slots, int N, int C, int H, int W, int groups, float eps, float momentum, int relu, void* stream);

/* Batched weight re-layout.  A training step re-derives the kernel-side copy of every conv weight once per optimiser
 * step; instead of one small launch per convolution inside fd_conv2d_fwd / fd_conv2d_bwd_data (wt_ready = 0) the caller can
 * collect the work of all its convolutions once and run it as ONE launch after each optimiser update, then call the
 * convolutions with wt_ready = 1.
 *   fd_conv2d_relayout_jobs  appends to jobs[] (capacity >= 4) what the forward (kind 0) or data-gradient (kind 1) of `d`
 *                            would re-lay-out from `w` into `wt`; returns the number of jobs (0: that path needs no copy).
 *   fd_relayout_plan         fills the launch bookkeeping of a host array of n jobs; returns the workgroup count.
 *   fd_relayout_batch        runs n jobs; `jobs_dev` is the planned array copied to DEVICE memory. */
typedef struct fd_relayout_job {
    const float* w;
    float* dst;
    int Co, Ci, KH, KW, TA, TB, kh0, dkh, kw0, dkw;
    int mode;          /* 0 forward [Co][tap][Ci], 1 data-gradient [Ci][tap][Co], 2 generic data-gradient [Ci][Co][tap]; 3 - 6 Winograd U;
                          7 / 8 the 1x1 matrix, 9 / 10 the [m][(tap, channel)] matrix of layouts 0 / 1 pre-split into bf16 limbs (csrc/conv_limb.h); 11 / 12 the limb image of layouts 5 / 6 */
    int reserved;
    long n;            /* elements */
    long first_block;  /* set by fd_relayout_plan */
} fd_relayout_job;
int fd_conv2d_relayout_jobs(const fd_conv_desc* d, int kind, const float* w, float* wt, fd_relayout_job* jobs);
long fd_relayout_plan(fd_relayout_job* jobs_host, int n);
int fd_relayout_batch(const fd_relayout_job* jobs_dev, int n, long total_blocks, void* stream);

/* gx [N,Cin,H,W] = d/dx of sum(conv(x,w) * gy)  (gy is the gradient w.r.t. the PRE-activation output;
 * apply fd_act_bwd first when act != 0).  wt / wt_ready as in fd_conv2d_fwd (flipped / transposed layouts, one per
 * output-parity class for stride 2: fd_conv2d_bwd_data_wt_floats(d) floats); ws: fd_conv2d_bwd_data_ws_floats(d). */
long fd_conv2d_bwd_data_wt_floats(const fd_conv_desc* d);
long fd_conv2d_bwd_data_ws_floats(const fd_conv_desc* d);
int fd_conv2d_bwd_data(const fd_conv_desc* d, const float* gy, const float* w, float* gx, float* wt, int wt_ready, float* ws,
                       void* stream);
/* gx = (data gradient as above) + gx_add, gx_add [N,Cin,H,W] (must not alias gx): where a tensor feeds a convolution AND a second
 * consumer - the input of a ResNet block also is its residual branch (torchvision BasicBlock / Bottleneck `out += identity`, used
 * by networks/resnet_encoder.py:61-75) - autograd has to sum two gradients; here the second one joins in the epilogue of the
 * data-gradient kernel instead of costing an element-wise pass over both tensors.  Bitwise the sum torch would form. */
int fd_conv2d_bwd_data_add(const fd_conv_desc* d, const float* gy, const float* w, const float* gx_add, float* gx, float* wt,
                           int wt_ready, float* ws, void* stream);
/* gx = (data gradient as above) * act'(x_in): for a layer whose INPUT x_in is the output of activation in_act (1 ReLU, 2 ELU, 3 sigmoid,
 * 4 tanh) and is consumed by this layer alone, the result is the gradient w.r.t. the producer's PRE-activation - what fd_act_bwd
 * would compute from gx in a separate pass over the tensor (layers.py:100-112 ConvBlock -> networks/depth_decoder.py:92 dispconv:
 * the full-resolution ELU output feeds the disparity head only).  Fused into the store of the one-output-channel stencil; the other
 * kernel families run the data gradient followed by the element-wise pass (same result). */
int fd_conv2d_bwd_data_inact(const fd_conv_desc* d, const float* gy, const float* w, const float* x_in, int in_act, float* gx, float* wt,
                             int wt_ready, float* ws, void* stream);
/* gw [Cout,Cin,KH,KW], gbias [Cout] (NULL to skip).  accumulate != 0: gw += / gbias += (gradient accumulation straight
 * into the caller's buffers).  ws: fd_conv2d_bwd_weight_ws_floats(d) floats. */
long fd_conv2d_bwd_weight_ws_floats(const fd_conv_desc* d);
int fd_conv2d_bwd_weight(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* gbias, float* ws,
                         int accumulate, void* stream);

/* gpre = gy * act'(y)  where y is the activation OUTPUT (1 ReLU, 2 ELU(alpha=1), 3 sigmoid, 4 tanh). */
/* Convolution + training-mode BatchNorm (+ residual, ReLU) of a deep ResNet block in two launches (round 5): for the 3x3 layers that
 * run as F(2x2, 3x3) slabs (layer3 / layer4) the slab reduction of the convolution happens inside the small-plane BatchNorm kernel.
 * fd_conv2d_fwd_bn_ok(d, groups) == 1 says that this convolution (no bias, no activation) + BatchNorm qualifies; then
 * fd_conv2d_fwd_bn == fd_conv2d_fwd(d, x, w, NULL, y, wt, wt_ready, ws) followed by fd_bn_train_fwd(y, ..., out, ...): same y bit
 * for bit, same statistics rule as the small-plane kernel.  ws / wt as for fd_conv2d_fwd. */
int fd_conv2d_fwd_bn_ok(const fd_conv_desc* d, int groups);
int fd_conv2d_fwd_bn(const fd_conv_desc* d, const float* x, const float* w, float* y, float* wt, int wt_ready, float* ws,
                     const float* bn_weight, const float* bn_bias, const float* residual, float* out, float* running_mean,
                     float* running_var, float* save_mean, float* save_invstd, int groups, float eps, float momentum, int relu,
                     void* stream);
int fd_act_bwd(const float* y, const float* gy, float* gpre, long n, int act, void* stream);

/* nn.BatchNorm2d in training mode (torchvision ResNet; trainer.py:207-211 set_train), optionally fused with the
 * residual add and ReLU of a BasicBlock/Bottleneck tail:  y = relu?( bn(x) + residual? ).
 * `groups` G >= 1 splits the batch into G consecutive sub-batches that are normalised independently — bit-for-bit what G
 * separate forward passes would do (predict_poses runs the pose encoders once per source frame, trainer.py:336-351; the
 * trainer batches those passes) — including G in-order momentum updates of the running statistics.
 * save_mean / save_invstd [G*C] are written for the backward; running_mean / running_var are updated in place
 * with `momentum` (unbiased variance), as torch does.  ws: fd_bn_ws_floats(N,C,H,W,G). */
long fd_bn_ws_floats(int N, int C, int H, int W, int groups);
int fd_bn_train_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* ws, int N, int C,
                    int H, int W, int groups, float eps, float momentum, int relu, void* stream);
/* eval-mode BN (running statistics), same fusion options. */
int fd_bn_eval_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                   const float* running_mean, const float* running_var, int N, int C, int H, int W, float eps, int relu,
                   void* stream);
/* Backward of the fused op.  y = forward output (for the ReLU mask when relu=1).  gx, gweight, gbias are
 * written; g_residual (if non-NULL) receives the masked upstream gradient (the residual branch's gradient). */
int fd_bn_train_bwd(const float* x, const float* y, const float* gy, const float* weight, const float* save_mean,
                    const float* save_invstd, float* gx, float* gweight, float* gbias, float* g_residual, float* ws, int N,
                    int C, int H, int W, int groups, int relu, int accumulate /* gweight/gbias += */, void* stream);

/* The same backward for a BatchNorm + ReLU WITHOUT a residual input (bn1 of a BasicBlock, bn1 / bn2 of a Bottleneck): the ReLU mask
 * is recomputed from x (weight * invstd * (x - mean) + bias > 0), so the forward output is not read - two tensors per pass instead
 * of three (round 5). */
int fd_bn_train_bwd_remask(const float* x, const float* gy, const float* weight, const float* bias, const float* save_mean,
                           const float* save_invstd, float* gx, float* gweight, float* gbias, float* ws, int N, int C, int H, int W,
                           int groups, int accumulate, void* stream);

/* nn.MaxPool2d(3, stride 2, padding 1) (resnet_encoder.py:98).  idx [N,C,Ho,Wo] u8 = argmax tap (0..8). */
int fd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int C, int H, int W, void* stream);
int fd_maxpool3x3s2_bwd(const float* gy, const uint8_t* idx, float* gx, int N, int C, int H, int W, void* stream);

/* The stem's tail in one pass (resnet_encoder.py:95-98: features[0] = relu(bn1(conv1(x))), x = maxpool(features[0])), training mode:
 * fd_bn_train_fwd(relu = 1) + fd_maxpool3x3s2_fwd without the round trip of the full-resolution activation through memory.
 *   fwd  x [N,C,H,W] (the convolution's output) -> pooled [N,C,Ho,Wo], idx (argmax tap, u8), and - only when feat != NULL - feat
 *        [N,C,H,W] = relu(bn(x)) (the encoders whose features[0] nobody reads, the pose encoders, pass NULL); statistics, groups,
 *        running statistics, save_mean / save_invstd and ws exactly as fd_bn_train_fwd.
 *   bwd  g_pooled (+ g_feat, the gradient arriving at features[0] from its other consumer, or NULL) -> gx [N,C,H,W], gweight, gbias:
 *        the ReLU mask and the normalised value are recomputed from x, the pooling adjoint from g_pooled and idx. */
int fd_bn_relu_maxpool_fwd(const float* x, const float* weight, const float* bias, float* feat, float* pooled, uint8_t* idx,
                           float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* ws, int N, int C, int H,
                           int W, int groups, float eps, float momentum, void* stream);
int fd_bn_relu_maxpool_bwd(const float* x, const float* g_pooled, const uint8_t* idx, const float* g_feat, const float* weight,
                           const float* bias, const float* save_mean, const float* save_invstd, float* gx, float* gweight, float* gbias,
                           float* ws, int N, int C, int H, int W, int groups, int accumulate, void* stream);

/* Decoder input assembly (networks/depth_decoder.py:75-83): out = cat([nearest_up2(a), s1 (+ s2), s3], dim=1).
 * a [N,Ca,h,w] -> channels [0,Ca) at (2h,2w); s1,s2 [N,Cs,2h,2w] (s2 optional addend: beam-feature fusion
 * depth_decoder.py:78); s3 [N,C3,2h,2w] optional (refiner depth_maps, :81-82).  Any of s1/s3 may be NULL. */
int fd_upcat_fwd(const float* a, const float* s1, const float* s2, const float* s3, float* out, int N, int Ca, int Cs,
                 int C3, int h, int w, void* stream);
/* ga [N,Ca,h,w] (2x2 sums), gs [N,Cs,2h,2w] (shared by s1 and s2), g3 [N,C3,2h,2w]; each may be NULL. */
int fd_upcat_bwd(const float* gout, float* ga, float* gs, float* g3, int N, int Ca, int Cs, int C3, int h, int w,
                 void* stream);
/* The same with ga *= act'(a_out): `a` was the output of activation a_act (layers.py:100-112 ConvBlock, upconv(i, 0)) and feeds this
 * concatenation only, so ga is the gradient w.r.t. that layer's PRE-activation (no separate fd_act_bwd pass). */
int fd_upcat_bwd_act(const float* gout, const float* a_out, int a_act, float* ga, float* gs, float* g3, int N, int Ca, int Cs, int C3,
                     int h, int w, void* stream);
/* layers.py:229-232 upsample (nearest x2) alone */
int fd_upsample2x_fwd(const float* x, float* y, long planes, int h, int w, void* stream);
int fd_upsample2x_bwd(const float* gy, float* gx, long planes, int h, int w, void* stream);

/* trainer.py:569-596  loss_s = photo_s + w * smooth_s / 2^s ;  total = (sum_s loss_s + sum_s si_s) / n_scales  on device
 * scalars (photo / smooth / si: n_scales host arrays of device pointers to one float; si[s] may be NULL).
 * out[0..n-1] = loss_s, out[n] = total.  bwd: grads[0..n-1] = d total/d photo_s, [n..2n-1] = d/d smooth_s, [2n..3n-1] = d/d si_s. */
int fd_combine_losses_fwd(const float* const* photo, const float* const* smooth, const float* const* si, int n_scales,
                          float smooth_weight, float* out, void* stream);
int fd_combine_losses_bwd(const float* g_total, int n_scales, float smooth_weight, float* grads, void* stream);

/* networks/resnet_encoder.py:94  y = (x - mean) / std  elementwise (the encoder's input normalisation, 0.45 / 0.225).
 * The same arithmetic is available fused into the stem conv through fd_conv_desc.in_norm. */
int fd_input_normalize(const float* x, float* y, long n, float mean, float std, void* stream);
/* The pose networks' input in one pass (trainer.py:336-351 `torch.cat([inputs[("color_aug", f_i, 0)], inputs[("color_aug", f_j, 0)]], 1)`
 * for every source frame, stacked along the batch axis, followed by resnet_encoder.py:94): piece p copies `imgs` whole images
 * [imgs][C][H][W] from src[p] (device pointers, array on the HOST) to images dst_img[p] .. dst_img[p] + imgs - 1 of out
 * [n][Ct][H][W] at channel offset dst_ch[p], optionally as (x - mean) / std.  At most 16 pieces per call. */
int fd_stack_normalize(const float* const* src, const int* dst_img, const int* dst_ch, int n_pieces, int imgs, int C, int Ct, int H,
                       int W, float* out, int normalize, float mean, float std, void* stream);

/* out = a + b (feature fusion depth_decoder.py:70, pose_decoder.py:31) ; out = alpha*a + beta*b */
int fd_axpby(const float* a, const float* b, float* out, long n, float alpha, float beta, void* stream);

/* pose_decoder.py:44-46: out[n][c] = scale * mean over (H,W) of x[n][c]; and its adjoint. */
int fd_spatial_mean_fwd(const float* x, float* out, long planes, long plane_size, float scale, void* stream);
int fd_spatial_mean_bwd(const float* gout, float* gx, long planes, long plane_size, float scale, void* stream);

/* layers.py:284-302 compute_depth_errors on n matched (gt, pred) values: out[7] =
 * abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3.  ws: 7*256 floats. */
int fd_depth_errors(const float* gt, const float* pred, long n, float* out, float* ws, void* stream);

/* evaluate_depth.py:62-70 batch_post_process_disparity(l_disp, r_disp): l_disp / r_disp [planes][H][W] float32 (r_disp = the
 * prediction for the mirrored image, already flipped back) -> out [planes][H][W] float64, bit-exact vs the numpy expression. */
int fd_post_process_disparity(const float* l_disp, const float* r_disp, double* out, long planes, int H, int W, void* stream);

/* ------------------------------------------------------------------ optimiser ------------------ */

/* torch.optim.Adam step (trainer.py:129,247) over a flat parameter segment:
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; p -= step_size * m / (sqrt(v)/sqrt(bc2) + eps)
 * with step_size = lr/bc1, bc1 = 1-b1^t, bc2 = 1-b2^t computed by the caller.  grad_scale multiplies g first
 * (1/world_size averaging for data parallel).  */
int fd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr, float beta1,
                 float beta2, float eps, float bias_corr1, float bias_corr2, float grad_scale, void* stream);
/* Same update with the step counter and learning rate held on the device (state[0] = step count as float, incremented
 * by this call; state[1] = lr), so that a captured hipGraph of the training step stays valid across replays. */
int fd_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float* state, float beta1,
                     float beta2, float eps, float grad_scale, void* stream);

/* ------------------------------------------------------------------ sparse LiDAR --------------- */

/* gen2channel.py:60-117 get_4beam_2channel, gather formulation (race-free, bit-exact vs the
 * sequential reference): beam [B,1,H,W] -> out [B,2,H,W] (ch0 expanded depth, ch1 confidence).
 * Donor ROI rows [r0,r1), cols [c0,c1) (76,190,2,638 for 192x640); expand = 2. */
int fd_scatter_2channel(const float* beam, float* out, int B, int H, int W, int r0, int r1, int c0, int c1, int expand,
                        void* stream);

/* 3x3 stride-1 pad-1 convolution (nn.Conv2d / layers.Conv3x3 as above) through the 1-D Winograd F(2,3) transform: 1.5x fewer
 * MFMA cycles than the direct implicit GEMM.  Needs Cin % 16 == 0 and an even width (fd_conv3x3_wino_wt_floats returns 0
 * otherwise).  `wt` receives the transformed weights [4][Cout][3][Cin]; pass wt_ready = 1 to reuse them.  fd_conv2d_fwd routes
 * eligible convolutions here by itself; these entry points expose the path on its own (probes, tests). */
long fd_conv3x3_wino_wt_floats(const fd_conv_desc* d);
long fd_conv3x3_wino_ws_floats(const fd_conv_desc* d);
int fd_conv3x3_wino_fwd(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt, int wt_ready,
                        float* ws, void* stream);

/* LiDAR rasterisation (the step upstream of the scatter): kitti_utils.py:40-102 generate_depth_map + kitti_dataset.py:93-117
 * get_4beam + mono_dataset.py:193-198.  points [n][4] float32 (forward, left, up, reflectance), P_velo2im 3x4 float64 (device),
 * image im_h x im_w  ->  z-buffered sparse depth (float64; vel_depth != 0 stores the forward distance instead of the camera z,
 * kitti_utils.py:68-69), padded / cropped to target_h x target_w exactly as generate_depth_map(shape=[target_h, target_w]) does
 * (target_h <= 0: shape=None).  Outputs, either may be NULL:
 *   depth_out [padded_h][target_w] float64 = the function's return value  (padded_h = im_h + |target_h - im_h| - (2 if target_h < im_h))
 *   beam_out  [ceil(padded_h / 2)][ceil(target_w / 2)] float32 = 2x2 ceil-mode max-pool of it / 100 = the "4beam" network input.
 * Bit-exact vs the reference incl. its duplicate rule (oracle/rasterize.py).  ws: fd_velo_rasterize_ws_bytes bytes. */
long fd_velo_rasterize_ws_bytes(int n_points, int im_h, int im_w);
int fd_velo_rasterize(const float* points, int n_points, const double* P_velo2im, int im_h, int im_w, int vel_depth, int target_h,
                      int target_w, float* beam_out, double* depth_out, void* ws, void* stream);

/* evaluate_depth.py:349 `cv2.resize(pred_disp, (gt_width, gt_height))` on a float32 image (default INTER_LINEAR): OpenCV's
 * coefficient rule (source coordinate (float)((d + 0.5) * scale - 0.5) with the scale in double, floor, edges clamp with weight 0)
 * and its two float32 passes, horizontal first.  x [planes][Hin][Win] -> y [planes][Hout][Wout].  Restated from OpenCV's source
 * (resize.cpp); OpenCV is not available in the build image, so this entry point is parity-UNPINNED (oracle/evaluate.py). */
int fd_resize_linear_cv(const float* x, float* y, long planes, int Hin, int Win, int Hout, int Wout, void* stream);

/* ------------------------------------------------------------------ refiner inputs ------------- */

/* torch.median(x[mask] * scale) with mask = gate > 0 inside rows [y0,y1) x columns [x0,x1) of every [H][W] plane of the batch
 * (refiner.py:327-331: `torch.median(beam[mask] * 100.0)`): the LOWER median (rank (n-1)/2) over the whole batch, by radix select on
 * order-preserving integer keys - no sort, no host round trip for the selection size, result independent of the order in which
 * the selection was compacted.  out[0] = median (NaN when the selection is empty - the reference raises there - or holds a NaN),
 * out[1] = n.  x and gate: [B][H][W].  ws: fd_masked_median_ws_bytes(B, H, W) bytes. */
long fd_masked_median_ws_bytes(int B, int H, int W);
int fd_masked_median(const float* x, const float* gate, float scale, int B, int H, int W, int y0, int y1, int x0, int x1, float* out,
                     void* ws, void* stream);

/* refiner.py:316-348: the depth-map inputs of the refine decoder for ALL scales, written straight into the concatenated tensors
 * out[s] [B][1 + 3*catxy + 2][Hs][Ws] = (scaled_disp | Cat_xy(s-fold max-pooled depth, inv_K[s]) | s-fold max-pooled 2-channel map):
 * per scale, disp[s] (or, pool_disp0 != 0 = --refine_a0 true, the s-fold 2x2 ceil-mode max-pool of disp[0]) is up-sampled bilinearly
 * to H x W, turned into depth, scaled by median(beam[mask] * 100) / median(depth[mask]) (mask: beam > 0 inside the crop, medians over
 * the batch, fd_masked_median's rule), and scaled_disp = (bilinear down-sample of 1 / depth - 0.01) / 9.9.  Four launches, no
 * full-resolution intermediate.  Hs[s] * r == H, Ws[s] * r == W with r a power of two <= 8.  No gradient (the reference runs this
 * under no_grad / detaches the ratio; the decoder's inputs are leaves).  stats (may be NULL): [n_scales][4] = ratio,
 * median(depth[mask]), median(beam[mask] * 100), n.  ws: fd_refine_inputs_ws_bytes(cfg) bytes. */
typedef struct fd_refine_cfg {
    int B, H, W, n_scales;
    int Hs[4], Ws[4];
    int crop_y0, crop_y1, crop_x0, crop_x1;     /* refiner.py:329: rows 78..189, columns 23..616 */
    double min_depth, max_depth;
    int catxy, pool_disp0;
} fd_refine_cfg;
long fd_refine_inputs_ws_bytes(const fd_refine_cfg* cfg);
int fd_refine_inputs(const fd_refine_cfg* cfg, const float* const* disp, const float* beam, const float* two_cha,
                     const float* const* inv_K, float* const* out, float* stats, void* ws, void* stream);

/* ------------------------------------------------------------------ replay (round 6) ------------
 * Issue a recorded sequence of entry-point calls from ONE host call.  The Python layer pays 10 - 18 us of interpreter / ctypes /
 * autograd time per launch; the parts of a step that repeat the same calls on the same shapes and need no autograd graph - the
 * frozen stage-1 networks of refiner.py:299-330, validation forwards (trainer.py:390-423) - are recorded once and replayed here.
 *   fd_call_rec      one call: `fn` = index into the table of stream-ordered entry points (fd_replay_function_name / _signature:
 *                    every int-status entry point of this header whose last argument is the stream), `arg[i]` its arguments as
 *                    64-bit words (float / double arguments: the bits of a double), `kind[i]` & 15 = how arg[i] becomes the value:
 *                    0 literal, 1 arena + arg[i] bytes, 2 inputs[kind[i] >> 4] + arg[i] bytes, 3 the stream passed to fd_replay.
 *   fd_replay        runs the records in order on `stream`; stops at the first non-zero status and returns it.  Nothing is allocated
 *                    or synchronised: the caller owns `arena` (the records' intermediates and outputs live in it), the tensors
 *                    behind literal pointers (parameters, cached weight layouts, descriptors) and the inputs. */
#define FD_REPLAY_MAX_ARGS 24
typedef struct fd_call_rec {
    int fn, nargs;
    long long arg[FD_REPLAY_MAX_ARGS];
    unsigned short kind[FD_REPLAY_MAX_ARGS];
} fd_call_rec;
int fd_replay_function_count(void);
const char* fd_replay_function_name(int i);
const char* fd_replay_function_signature(int i);
int fd_replay(const fd_call_rec* recs, int n, void* arena, const void* const* inputs, int n_inputs, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FDHIP_H */
