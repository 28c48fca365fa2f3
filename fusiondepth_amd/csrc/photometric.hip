// Fused photometric-reprojection + sparse-LiDAR loss for one pyramid scale (forward and backward),
// plus stand-alone SSIM / reprojection-loss-map kernels for the layers.py API.
//
// Reference path replaced (per scale): trainer.py:434-470 (bilinear upsample of disp, disp_to_depth,
// BackprojectDepth, Project3D, F.grid_sample border/bilinear/align_corners=False) and
// trainer.py:476-488,509-567,577-589 (SSIM+L1, identity losses + noise, per-pixel min, mean, masked
// scale-invariant log loss).  The reference materialises ~40 intermediate tensors per scale; here
// one workgroup owns a 16x64 pixel tile, stages the warped prediction and the target (with their
// reflect-pad halo) in LDS and produces the per-tile partial sums directly.  HBM-bound by design:
// compulsory traffic per scale is disp_s + target + 2 sources + LiDAR (+ 1 byte/pixel argmin).
//
// Thread layout: NT threads = NT/64 waves; thread (tx = tid&63, ty = tid>>6) owns the PPT = 16/(NT/64) vertically
// adjacent pixels (ty*PPT .. ty*PPT+PPT-1, tx) of the tile, so lanes of a wave touch consecutive x (coalesced
// HBM rows, conflict-free LDS rows) and the 3x3 box sums slide down the column.  The kernels are bound by the vector
// ALU and by the latency of their dependent gathers (profiles/round1_pmc_loss.md); 8 waves x 2 pixels per tile keeps 24
// waves per CU resident against 12 with 4 x 4.
#include "../../include/fdhip.h"
#include "fd_common.h"

namespace {

constexpr int TW = 64, TH = 16;
#ifndef FD_PHOTO_NT
#define FD_PHOTO_NT 512      // measured at batch 12: 256 -> 65 / 209 us (fwd / bwd), 512 -> 61 / 173, 1024 -> 87 / 214
#endif
constexpr int NT = FD_PHOTO_NT;          // threads per 16x64 tile
constexpr int NWV = NT / 64;              // waves per tile
constexpr int PPT = TH / NWV;             // vertically adjacent pixels owned by a thread (4 at 256 threads, 2 at 512)
constexpr int W1 = TW + 2, H1 = TH + 2;  // halo-1 region (SSIM window of tile pixels)
constexpr int W2 = TW + 4, H2 = TH + 4;  // halo-2 region (SSIM windows of halo-1 pixels, backward)
constexpr float C1 = (float)(0.01 * 0.01), C2 = (float)(0.03 * 0.03);

// The kernels below are VALU-bound (profiles/round1_pmc_loss.md) and an IEEE fp32 division expands to ~10 instructions.
// fdiv: v_rcp_f32 + one Newton-Markstein correction with FMAs (4 instructions) - the correctly rounded quotient except
// in rare last-bit cases.  div9 / div3: division by a constant as multiply + FMA residual + FMA correction, which
// reproduces IEEE division bit for bit (checked on 8M samples) - the window means and E[x^2] - mu^2 variances of the SSIM
// cancel catastrophically, so their divisions by 9 must round exactly as the reference's AvgPool2d does.
__device__ __forceinline__ float fdiv(float a, float b) {
    const float rc = __builtin_amdgcn_rcpf(b);
    const float q0 = a * rc;
    return fmaf(fmaf(-q0, b, a), rc, q0);
}
__device__ __forceinline__ float frcp(float x) { return fdiv(1.0f, x); }
__device__ __forceinline__ float div9(float x) {
    const float q0 = x * (1.0f / 9.0f);
    return fmaf(fmaf(-q0, 9.0f, x), 1.0f / 9.0f, q0);
}
__device__ __forceinline__ float div3(float x) {
    const float q0 = x * (1.0f / 3.0f);
    return fmaf(fmaf(-q0, 3.0f, x), 1.0f / 3.0f, q0);
}

__device__ __forceinline__ int refl_clamp(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return fd_clampi(i, 0, n - 1);
}

struct Cam {
    float ik[9];   // inv_K[:3,:3]
    float lo, span;
    float sh, sw;  // Hs/H, Ws/W
    float eps;
    float rw1, rh1;   // 1/(W-1), 1/(H-1)
    int H, W, Hs, Ws;
};

__device__ __forceinline__ Cam make_cam(const fd_photo_cfg& c, const float* invK_b) {
    Cam cm;
    cm.ik[0] = invK_b[0]; cm.ik[1] = invK_b[1]; cm.ik[2] = invK_b[2];
    cm.ik[3] = invK_b[4]; cm.ik[4] = invK_b[5]; cm.ik[5] = invK_b[6];
    cm.ik[6] = invK_b[8]; cm.ik[7] = invK_b[9]; cm.ik[8] = invK_b[10];
    cm.lo = (float)(1.0 / c.max_depth);
    cm.span = (float)(1.0 / c.min_depth - 1.0 / c.max_depth);
    cm.sh = (float)c.Hs / (float)c.H;
    cm.sw = (float)c.Ws / (float)c.W;
    cm.eps = c.eps;
    cm.rw1 = 1.0f / (float)(c.W - 1); cm.rh1 = 1.0f / (float)(c.H - 1);     // used for gradients only (see project)
    cm.H = c.H; cm.W = c.W; cm.Hs = c.Hs; cm.Ws = c.Ws;
    return cm;
}

// disp_s bilinearly resized to (H,W) at pixel (y,x)   (trainer.py:434-435)
__device__ __forceinline__ float disp_up_at(const float* __restrict__ d, const Cam& cm, int y, int x) {
    int y0, y1, x0, x1;
    float ly, lx;
    fd_bilinear_src(y, cm.sh, cm.Hs, y0, y1, ly);
    fd_bilinear_src(x, cm.sw, cm.Ws, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* r0 = d + (long)y0 * cm.Ws;
    const float* r1 = d + (long)y1 * cm.Ws;
    return hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
}

struct Samp {      // grid_sample(border, bilinear, align_corners=False) source position
    float gx, gy;  // normalised grid (the reference's ("sample", f, s))
    float ix, iy;  // clipped source coordinates
    float u, v, den, rden;
    float X[3];    // camera point
    float ray[3];
    float depth;
    int x0, y0;
    float fx, fy;       // ix - x0, iy - y0
    float mx, my;       // d ix / d gx (0 when clipped), d iy / d gy
};

__device__ __forceinline__ void project(const Cam& cm, const float* __restrict__ P, float dup, int y, int x, Samp& s) {
    const float fx = (float)x, fy = (float)y;
    s.depth = frcp(cm.lo + cm.span * dup);                          // layers.py:18-19
    s.ray[0] = cm.ik[0] * fx + cm.ik[1] * fy + cm.ik[2];            // layers.py:158
    s.ray[1] = cm.ik[3] * fx + cm.ik[4] * fy + cm.ik[5];
    s.ray[2] = cm.ik[6] * fx + cm.ik[7] * fy + cm.ik[8];
    s.X[0] = s.depth * s.ray[0]; s.X[1] = s.depth * s.ray[1]; s.X[2] = s.depth * s.ray[2];
    const float c0 = P[0] * s.X[0] + P[1] * s.X[1] + P[2] * s.X[2] + P[3];   // layers.py:219
    const float c1 = P[4] * s.X[0] + P[5] * s.X[1] + P[6] * s.X[2] + P[7];
    const float c2 = P[8] * s.X[0] + P[9] * s.X[1] + P[10] * s.X[2] + P[11];
    s.den = c2 + cm.eps;
    s.rden = frcp(s.den);
    s.u = fdiv(c0, s.den);                                          // layers.py:221
    s.v = fdiv(c1, s.den);
    s.gx = (fdiv(s.u, (float)(cm.W - 1)) - 0.5f) * 2.0f;            // layers.py:224-226
    s.gy = (fdiv(s.v, (float)(cm.H - 1)) - 0.5f) * 2.0f;
    // aten GridSampler.h: unnormalize (align_corners=False) then clip_coordinates (border)
    float ix = ((s.gx + 1.0f) * (float)cm.W - 1.0f) * 0.5f;
    float iy = ((s.gy + 1.0f) * (float)cm.H - 1.0f) * 0.5f;
    const float xm = (float)(cm.W - 1), ym = (float)(cm.H - 1);
    s.mx = (ix <= 0.f || ix >= xm) ? 0.f : (float)cm.W * 0.5f;      // clip_coordinates_set_grad
    s.my = (iy <= 0.f || iy >= ym) ? 0.f : (float)cm.H * 0.5f;
    ix = fminf(xm, fmaxf(ix, 0.f));
    iy = fminf(ym, fmaxf(iy, 0.f));
    s.ix = ix; s.iy = iy;
    const float flx = floorf(ix), fly = floorf(iy);
    s.x0 = (int)flx; s.y0 = (int)fly;
    s.fx = ix - flx; s.fy = iy - fly;
}

// 4-tap gather of the three colour planes; taps outside the image contribute 0 (only possible for
// the +1 taps when ix == W-1 / iy == H-1, where their weight is 0 anyway).
__device__ __forceinline__ void gather3(const float* __restrict__ src, const Cam& cm, const Samp& s, float (&out)[3],
                                        float (&taps)[3][4]) {
    const long P = (long)cm.H * cm.W;
    const bool xin = s.x0 + 1 <= cm.W - 1, yin = s.y0 + 1 <= cm.H - 1;
    const long o00 = (long)s.y0 * cm.W + s.x0;
    const float wx1 = s.fx, wx0 = 1.f - s.fx, wy1 = s.fy, wy0 = 1.f - s.fy;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = src + c * P + o00;
        const float nw = p[0];
        const float ne = xin ? p[1] : 0.f;
        const float sw = yin ? p[cm.W] : 0.f;
        const float se = (xin && yin) ? p[cm.W + 1] : 0.f;
        taps[c][0] = nw; taps[c][1] = ne; taps[c][2] = sw; taps[c][3] = se;
        out[c] = nw * (wy0 * wx0) + ne * (wy0 * wx1) + sw * (wy1 * wx0) + se * (wy1 * wx1);
    }
}

// SSIM loss value from the five 3x3 window sums (layers.py:267-281).
__device__ __forceinline__ float ssim_from_sums(float Sx, float Sy, float Sxx, float Syy, float Sxy) {
    const float mx = div9(Sx), my = div9(Sy);
    const float sx = div9(Sxx) - mx * mx, sy = div9(Syy) - my * my, sxy = div9(Sxy) - mx * my;
    const float n = (2.f * mx * my + C1) * (2.f * sxy + C2);
    const float d = (mx * mx + my * my + C1) * (sx + sy + C2);
    const float v = (1.f - fdiv(n, d)) * 0.5f;
    return fminf(fmaxf(v, 0.f), 1.f);
}

// d(SSIM loss)/d(mu_x, E[x^2], E[xy]) at one window; all zero where the clamp is active.
__device__ __forceinline__ void ssim_coefs(float Sx, float Sy, float Sxx, float Syy, float Sxy, float w, float& ca,
                                           float& cb, float& cc) {
    const float mx = div9(Sx), my = div9(Sy);
    const float sx = div9(Sxx) - mx * mx, sy = div9(Syy) - my * my, sxy = div9(Sxy) - mx * my;
    const float A1 = 2.f * mx * my + C1, A2 = 2.f * sxy + C2;
    const float B1 = mx * mx + my * my + C1, B2 = sx + sy + C2;
    const float n = A1 * A2, d = B1 * B2;
    const float rd = frcp(d);
    const float q = n * rd;
    const float v = (1.f - q) * 0.5f;
    if (!(v >= 0.f && v <= 1.f)) { ca = 0.f; cb = 0.f; cc = 0.f; return; }
    ca = w * (-my * (A2 - A1) + q * mx * (B2 - B1)) * rd;
    cb = w * q * 0.5f * frcp(B2);
    cc = -w * A1 * rd;
}

// Per-thread column pass: for the 4 owned pixels compute, per channel, the SSIM loss and |t-p| from
// LDS planes sx (prediction) / sy (target) that hold the tile with `HALO`-wide border
// (plane row stride WS).  Returns 0.85*mean_c ssim + 0.15*mean_c l1 (or mean_c l1).
template <int WS, int HALO, bool SSIM>
__device__ __forceinline__ void column_losses(const float* __restrict__ sx, const float* __restrict__ sy, int plane,
                                              int tx, int ty, float (&L)[PPT]) {
    float ss[PPT], l1[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) { ss[i] = 0.f; l1[i] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* px = sx + c * plane;
        const float* py = sy + c * plane;
        if (SSIM) {
            float hx[PPT + 2], hy[PPT + 2], hxx[PPT + 2], hyy[PPT + 2], hxy[PPT + 2];
#pragma unroll
            for (int r = 0; r < PPT + 2; ++r) {
                const int o = (ty * PPT + r + HALO - 1) * WS + tx + HALO - 1;
                const float x0 = px[o], x1 = px[o + 1], x2 = px[o + 2];
                const float y0 = py[o], y1 = py[o + 1], y2 = py[o + 2];
                hx[r] = x0 + x1 + x2; hy[r] = y0 + y1 + y2;
                hxx[r] = x0 * x0 + x1 * x1 + x2 * x2;
                hyy[r] = y0 * y0 + y1 * y1 + y2 * y2;
                hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
            }
#pragma unroll
            for (int i = 0; i < PPT; ++i)
                ss[i] += ssim_from_sums(hx[i] + hx[i + 1] + hx[i + 2], hy[i] + hy[i + 1] + hy[i + 2],
                                        hxx[i] + hxx[i + 1] + hxx[i + 2], hyy[i] + hyy[i + 1] + hyy[i + 2],
                                        hxy[i] + hxy[i + 1] + hxy[i + 2]);
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int o = (ty * PPT + i + HALO) * WS + tx + HALO;
            l1[i] += fabsf(py[o] - px[o]);
        }
    }
#pragma unroll
    for (int i = 0; i < PPT; ++i) L[i] = SSIM ? 0.85f * div3(ss[i]) + 0.15f * div3(l1[i]) : div3(l1[i]);
}

struct PhotoArgs {
    fd_photo_cfg cfg;
    const float* disp; const float* inv_K; const float* P;
    const float* src[3];
    const float* target; const float* ident; const float* noise; const float* beam;
    const float* mask;       // [B,NF,H,W] multiplies the reprojection losses (trainer.py:530-541, --predictive_mask) or NULL
    uint8_t* sel;
    float* depth_out; float* sample_out; float* color_out;
    float* reproj_out;       // [B,NF,H,W] the UNmasked reprojection losses (what d/d mask needs) or NULL
    float* ws;
};

// ------------------------------------------------------------------------------------------------
template <bool SSIM>
__global__ void __launch_bounds__(NT) k_photo_fwd(PhotoArgs a) {
    __shared__ float s_tgt[3 * H1 * W1];
    __shared__ float s_pred[3 * H1 * W1];
    __shared__ float s_red[NWV * 4];
    const fd_photo_cfg& cfg = a.cfg;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0t = blockIdx.x * TW, y0t = blockIdx.y * TH, b = blockIdx.z;
    const int H = cfg.H, W = cfg.W, NF = cfg.NF;
    const long P = (long)H * W;
    const Cam cm = make_cam(cfg, a.inv_K + b * 16);
    const float* disp_b = a.disp + (long)b * cfg.Hs * cfg.Ws;

    for (int i = tid; i < H1 * W1; i += NT) {
        const int hy = i / W1, hx = i - hy * W1;
        const int gy = refl_clamp(y0t - 1 + hy, H), gx = refl_clamp(x0t - 1 + hx, W);
        const float* t = a.target + (long)b * 3 * P + (long)gy * W + gx;
        s_tgt[i] = t[0]; s_tgt[H1 * W1 + i] = t[P]; s_tgt[2 * H1 * W1 + i] = t[2 * P];
    }

    float Lr[3][PPT];
#pragma unroll
    for (int f = 0; f < 3; ++f) {
        if (f >= NF) {   // NF is workgroup-uniform, so the barriers below stay convergent
#pragma unroll
            for (int i = 0; i < PPT; ++i) Lr[f][i] = 0.f;
            continue;
        }
        const float* Pf = a.P + ((long)b * NF + f) * 12;
        const float* src_b = a.src[f] + (long)b * 3 * P;
        // fixed trip count + full unroll: the gathers of all iterations are independent, so the compiler can keep
        // several pixels' loads in flight (the kernel is latency-bound, not bandwidth-bound, at 12-16 waves per CU)
#pragma unroll
        for (int it = 0; it < (H1 * W1 + NT - 1) / NT; ++it) {
            const int i = tid + it * NT;
            if (i >= H1 * W1) break;
            const int hy = i / W1, hx = i - hy * W1;
            const int ry = y0t - 1 + hy, rx = x0t - 1 + hx;
            const int gy = refl_clamp(ry, H), gx = refl_clamp(rx, W);
            Samp s;
            project(cm, Pf, disp_up_at(disp_b, cm, gy, gx), gy, gx, s);
            float pr[3], taps[3][4];
            gather3(src_b, cm, s, pr, taps);
            s_pred[i] = pr[0]; s_pred[H1 * W1 + i] = pr[1]; s_pred[2 * H1 * W1 + i] = pr[2];
            const bool own = hy >= 1 && hy <= TH && hx >= 1 && hx <= TW && ry < H && rx < W;
            if (own) {
                const long p = (long)ry * W + rx;
                if (a.depth_out && f == 0) a.depth_out[b * P + p] = s.depth;
                if (a.sample_out) {
                    float2 g2; g2.x = s.gx; g2.y = s.gy;
                    reinterpret_cast<float2*>(a.sample_out)[((long)f * cfg.B + b) * P + p] = g2;
                }
                if (a.color_out) {
                    float* co = a.color_out + ((long)f * cfg.B + b) * 3 * P + p;
                    co[0] = pr[0]; co[P] = pr[1]; co[2 * P] = pr[2];
                }
            }
        }
        __syncthreads();
        column_losses<W1, 1, SSIM>(s_pred, s_tgt, H1 * W1, tx, ty, Lr[f]);
        if (a.mask || a.reproj_out) {
#pragma unroll
            for (int i = 0; i < PPT; ++i) {
                const int y = y0t + ty * PPT + i, xx = x0t + tx;
                if (y >= H || xx >= W) continue;
                const long q = ((long)b * NF + f) * P + (long)y * W + xx;
                if (a.reproj_out) a.reproj_out[q] = Lr[f][i];
                if (a.mask) Lr[f][i] *= a.mask[q];
            }
        }
        __syncthreads();
    }

    // per-pixel min over cat(identity + noise, reprojection)   (trainer.py:549-567)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};  // sum(min), n_valid, sum(d), sum(d^2)
    const int NI = a.ident ? (cfg.avg_reprojection ? 1 : NF) : 0;
    const int x = x0t + tx;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int y = y0t + ty * PPT + i;
        if (y >= H || x >= W) continue;
        const long p = (long)y * W + x;
        float best = 0.f;
        int bi = 0;
        bool first = true;
        for (int k = 0; k < NI; ++k) {
            float v = a.ident[((long)b * NI + k) * P + p];
            if (a.noise) v += a.noise[((long)b * NI + k) * P + p] * 0.00001f;
            if (first || v < best) { best = v; bi = k; first = false; }
        }
        if (cfg.avg_reprojection && NF >= 2) {
            const float v = NF == 2 ? (Lr[0][i] + Lr[1][i]) * 0.5f : div3(Lr[0][i] + Lr[1][i] + Lr[2][i]);
            if (first || v < best) { best = v; bi = NI; first = false; }
        } else {
#pragma unroll
            for (int f = 0; f < 3; ++f) {
                if (f >= NF) continue;
                const float v = Lr[f][i];
                if (first || v < best) { best = v; bi = NI + f; first = false; }
            }
        }
        acc[0] += best;
        a.sel[b * P + p] = (uint8_t)bi;
        if (a.beam) {  // trainer.py:577-589
            const float depth = frcp(cm.lo + cm.span * disp_up_at(disp_b, cm, y, x)) * cfg.si_depth_scale;
            const float bd = a.beam[b * P + p] * cfg.si_beam_scale;
            if (cfg.si_mode == 1) {  // completor.py:718-723 (--completion_l1loss): masked L1, no |pred - beam| gate
                if (bd > cfg.si_lo && depth < 80.f && depth > cfg.si_lo) { acc[1] += 1.f; acc[2] += fabsf(depth - bd); }
            } else if (bd > cfg.si_lo && depth < 80.f && depth > cfg.si_lo && fabsf(depth - bd) < cfg.si_threshold) {
                const float d = logf(depth) - logf(bd);
                acc[1] += 1.f; acc[2] += d; acc[3] += d * d;
            }
        }
    }
    const float s = fd_block_sum_n<4, NWV>(acc, s_red);
    if (tid < 4) {
        const long blk = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.ws[blk * 4 + tid] = s;
    }
}

// one workgroup per group (+ the photometric mean over all groups in block 0): fixed-order reduction of the per-tile
// partials, then the scalar loss terms.  A "group" is a sub-batch whose SI-log loss is evaluated on its own — the
// accumulated micro-batches of one optimiser step run as one stacked batch (trainer.py:237-248 sums their losses).
//   out[0] = mean over ALL pixels of the min-reprojection loss     out[4] = mean over groups of si_loss_g
//   out[8 + 4g ..] = n_valid_g, mean(d)_g, var_g, si_loss_g
__global__ void __launch_bounds__(256) k_photo_finalize(const float* __restrict__ ws, int tiles_per_group, int groups,
                                                        float count, float si_var, int have_beam, int si_mode,
                                                        float* __restrict__ out) {
    __shared__ float s_red[4 * 4];
    __shared__ float tot[4];
    const int g = blockIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = threadIdx.x; i < tiles_per_group; i += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += ws[((long)g * tiles_per_group + i) * 4 + k];
    const float s = fd_block_sum_n<4, 4>(acc, s_red);
    if (threadIdx.x < 4) tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float n = tot[1];
        const float m1 = tot[2] / n, m2 = tot[3] / n;
        const float var = m2 - si_var * (m1 * m1);
        float* o = out + 8 + 4 * g;
        o[0] = n; o[1] = m1; o[2] = var;
        o[3] = have_beam ? (si_mode == 1 ? m1 * 0.001f : sqrtf(var) * 0.1f) : 0.f;
        out[8 + 4 * groups + g] = tot[0];           // per-group sum of the min-reprojection loss
    }
}
__global__ void k_photo_finalize2(float* __restrict__ out, int groups, float count) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sum = 0.f, si = 0.f;
    for (int g = 0; g < groups; ++g) { sum += out[8 + 4 * groups + g]; si += out[8 + 4 * g + 3]; }
    out[0] = sum / count;
    out[1] = out[8]; out[2] = out[9]; out[3] = out[10];      // group 0 statistics (back-compat for groups == 1)
    out[4] = si / (float)groups;
    out[5] = 0.f; out[6] = 0.f; out[7] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// Backward.  ws layout: [nblk][36] gP partials (3 frames x 12), then [B,H,W] d(disp upsampled).
struct PhotoBwdArgs {
    fd_photo_cfg cfg;
    const float* disp; const float* inv_K; const float* P;
    const float* src[3];
    const float* target; const float* beam;
    const float* mask;       // as in the forward, or NULL
    const uint8_t* sel;
    const float* stats; const float* g;
    float* d_up;      // [B,H,W]
    float* part;      // [nblk][36]
    int has_ident;
};

// Sum of an LDS coefficient plane over the (reflect-pad adjoint) windows that contain pixel (qy,qx).
__device__ __forceinline__ float fold_sum(const float* __restrict__ plane, int qy, int qx, int y0t, int x0t, int H,
                                          int W) {
    float acc = 0.f;
    for (int a = 0; a < 3; ++a) {
        int ry;
        if (a == 0) ry = qy;
        else if (a == 1) { if (qy != 1) continue; ry = -1; }
        else { if (qy != H - 2) continue; ry = H; }
        for (int bb = 0; bb < 3; ++bb) {
            int rx;
            if (bb == 0) rx = qx;
            else if (bb == 1) { if (qx != 1) continue; rx = -1; }
            else { if (qx != W - 2) continue; rx = W; }
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy) {
                const int ly = ry + dy - (y0t - 1);
                if (ly < 0 || ly >= H1) continue;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int lx = rx + dx - (x0t - 1);
                    if (lx < 0 || lx >= W1) continue;
                    acc += plane[ly * W1 + lx];
                }
            }
        }
    }
    return acc;
}

template <bool SSIM>
__global__ void __launch_bounds__(NT) k_photo_bwd(PhotoBwdArgs a) {
    __shared__ float s_tgt[3 * H2 * W2];
    __shared__ float s_pred[3 * H2 * W2];
    __shared__ float s_coef[3 * H1 * W1];
    __shared__ uint8_t s_sel[H1 * W1];
    __shared__ float s_mask[H1 * W1];
    __shared__ float s_red[NWV * 36];
    const fd_photo_cfg& cfg = a.cfg;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0t = blockIdx.x * TW, y0t = blockIdx.y * TH, b = blockIdx.z;
    const int H = cfg.H, W = cfg.W, NF = cfg.NF;
    const long P = (long)H * W;
    const Cam cm = make_cam(cfg, a.inv_K + b * 16);
    const float* disp_b = a.disp + (long)b * cfg.Hs * cfg.Ws;
    const float g_photo = a.g[0] / ((float)cfg.B * (float)H * (float)W);   // d/d(min value) of the mean
    const int NI = a.has_ident ? (cfg.avg_reprojection ? 1 : NF) : 0;
    const bool avg = cfg.avg_reprojection && NF >= 2;
    const float wfrm = avg ? (NF == 2 ? 0.5f : 1.0f / 3.0f) : 1.0f;
    const int x = x0t + tx;

    for (int i = tid; i < H2 * W2; i += NT) {
        const int hy = i / W2, hx = i - hy * W2;
        const int gy = refl_clamp(y0t - 2 + hy, H), gx = refl_clamp(x0t - 2 + hx, W);
        const float* t = a.target + (long)b * 3 * P + (long)gy * W + gx;
        s_tgt[i] = t[0]; s_tgt[H2 * W2 + i] = t[P]; s_tgt[2 * H2 * W2 + i] = t[2 * P];
    }
    for (int i = tid; i < H1 * W1; i += NT) {
        const int hy = i / W1, hx = i - hy * W1;
        const int py = y0t - 1 + hy, px = x0t - 1 + hx;
        s_sel[i] = (py >= 0 && py < H && px >= 0 && px < W) ? a.sel[b * P + (long)py * W + px] : (uint8_t)255;
    }

    float d_depth[PPT];
#pragma unroll
    for (int i = 0; i < PPT; ++i) d_depth[i] = 0.f;
    float gP0[12], gP1[12], gP2[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) { gP0[i] = 0.f; gP1[i] = 0.f; gP2[i] = 0.f; }

    // d(loss)/d(pred channel c) for the 4 owned pixels; `c` is a compile-time constant so that every
    // register array below is statically indexed.
    auto channel_grad = [&](const int c, const int my_sel, float (&dp)[PPT]) __attribute__((always_inline)) {
        const float* px = s_pred + c * H2 * W2;
        const float* py = s_tgt + c * H2 * W2;
        if (SSIM) {
            // SSIM derivative coefficients at every halo-1 position selected for this frame
            for (int i = tid; i < H1 * W1; i += NT) {
                float ca = 0.f, cb = 0.f, cc = 0.f;
                if (s_sel[i] == my_sel) {
                    const int hy = i / W1, hx = i - hy * W1;
                    float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Syy = 0.f, Sxy = 0.f;
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                        for (int dx = 0; dx < 3; ++dx) {
                            const float xv = px[(hy + dy) * W2 + hx + dx], yv = py[(hy + dy) * W2 + hx + dx];
                            Sx += xv; Sy += yv; Sxx += xv * xv; Syy += yv * yv; Sxy += xv * yv;
                        }
                    ssim_coefs(Sx, Sy, Sxx, Syy, Sxy, g_photo * wfrm * (0.85f / 3.0f) * s_mask[i], ca, cb, cc);
                }
                s_coef[i] = ca; s_coef[H1 * W1 + i] = cb; s_coef[2 * H1 * W1 + i] = cc;
            }
            __syncthreads();
        }
        // 3x3 box sums of the three coefficient planes for the 4 owned pixels: horizontal 3-sums of the 6 halo-1 rows
        // slide down the column (18 LDS reads per plane instead of 36).  Pixels on the second / second-to-last row or
        // column of the IMAGE also collect the windows reached through the reflect padding: rare, handled by fold_sum.
        float box[3][PPT];
        if (SSIM) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* pl = s_coef + k * H1 * W1 + (ty * PPT) * W1 + tx;
                float hs[PPT + 2];
#pragma unroll
                for (int r = 0; r < PPT + 2; ++r) hs[r] = pl[r * W1] + pl[r * W1 + 1] + pl[r * W1 + 2];
#pragma unroll
                for (int i = 0; i < PPT; ++i) box[k][i] = hs[i] + hs[i + 1] + hs[i + 2];
            }
        }
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int y = y0t + ty * PPT + i;
            float g = 0.f;
            if (y < H && x < W) {
                const int o = (ty * PPT + i + 2) * W2 + tx + 2;
                const float xv = px[o], yv = py[o];
                if (SSIM) {
                    float sa = box[0][i], sb = box[1][i], sc = box[2][i];
                    if (y == 1 || y == H - 2 || x == 1 || x == W - 2) {
                        sa = fold_sum(s_coef, y, x, y0t, x0t, H, W);
                        sb = fold_sum(s_coef + H1 * W1, y, x, y0t, x0t, H, W);
                        sc = fold_sum(s_coef + 2 * H1 * W1, y, x, y0t, x0t, H, W);
                    }
                    g = div9(sa + 2.f * xv * sb + yv * sc);
                }
                if (s_sel[(ty * PPT + i + 1) * W1 + tx + 1] == my_sel) {
                    const float df = yv - xv;  // |t - p|' w.r.t. p
                    const float sg = df > 0.f ? -1.f : (df < 0.f ? 1.f : 0.f);
                    g += sg * g_photo * wfrm * ((SSIM ? 0.15f : 1.0f) / 3.0f) * s_mask[(ty * PPT + i + 1) * W1 + tx + 1];
                }
            }
            dp[i] = g;
        }
        if (SSIM) __syncthreads();  // s_coef is rewritten for the next channel
    };

    auto frame_pass = [&](const int f, float (&gp)[12]) __attribute__((always_inline)) {
        const float* Pf = a.P + ((long)b * NF + f) * 12;
        const float* src_b = a.src[f] + (long)b * 3 * P;
        const int my_sel = avg ? NI : NI + f;
        __syncthreads();  // previous readers of s_pred / s_mask are done; s_tgt / s_sel are complete
        for (int i = tid; i < H1 * W1; i += NT) {
            const int hy = i / W1, hx = i - hy * W1;
            const int py = y0t - 1 + hy, px = x0t - 1 + hx;
            s_mask[i] = (a.mask && py >= 0 && py < H && px >= 0 && px < W) ? a.mask[((long)b * NF + f) * P + (long)py * W + px] : 1.f;
        }
#pragma unroll
        for (int it = 0; it < (H2 * W2 + NT - 1) / NT; ++it) {
            const int i = tid + it * NT;
            if (i >= H2 * W2) break;
            const int hy = i / W2, hx = i - hy * W2;
            const int gy = refl_clamp(y0t - 2 + hy, H), gx = refl_clamp(x0t - 2 + hx, W);
            Samp s;
            project(cm, Pf, disp_up_at(disp_b, cm, gy, gx), gy, gx, s);
            float pr[3], taps[3][4];
            gather3(src_b, cm, s, pr, taps);
            s_pred[i] = pr[0]; s_pred[H2 * W2 + i] = pr[1]; s_pred[2 * H2 * W2 + i] = pr[2];
        }
        __syncthreads();
        float dp0[PPT], dp1[PPT], dp2[PPT];
        channel_grad(0, my_sel, dp0);
        channel_grad(1, my_sel, dp1);
        channel_grad(2, my_sel, dp2);

        // grid_sample backward (w.r.t. the grid) -> projection -> depth, and gP
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int y = y0t + ty * PPT + i;
            if (y >= H || x >= W) continue;
            if (dp0[i] == 0.f && dp1[i] == 0.f && dp2[i] == 0.f) continue;
            Samp s;
            project(cm, Pf, disp_up_at(disp_b, cm, y, x), y, x, s);
            float pr[3], taps[3][4];
            gather3(src_b, cm, s, pr, taps);
            const float go[3] = {dp0[i], dp1[i], dp2[i]};
            float gix = 0.f, giy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float nw = taps[c][0], ne = taps[c][1], sw = taps[c][2], se = taps[c][3];
                gix += go[c] * ((ne - nw) * (1.f - s.fy) + (se - sw) * s.fy);
                giy += go[c] * ((sw - nw) * (1.f - s.fx) + (se - ne) * s.fx);
            }
            const float du = gix * s.mx * 2.0f * cm.rw1;
            const float dv = giy * s.my * 2.0f * cm.rh1;
            const float dc0 = du * s.rden, dc1 = dv * s.rden, dc2 = -(du * s.u + dv * s.v) * s.rden;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                gp[j] += dc0 * s.X[j]; gp[4 + j] += dc1 * s.X[j]; gp[8 + j] += dc2 * s.X[j];
            }
            gp[3] += dc0; gp[7] += dc1; gp[11] += dc2;
            float dd = 0.f;
#pragma unroll
            for (int j = 0; j < 3; ++j) dd += (Pf[j] * dc0 + Pf[4 + j] * dc1 + Pf[8 + j] * dc2) * s.ray[j];
            d_depth[i] += dd;
        }
    };

    frame_pass(0, gP0);
    if (NF > 1) frame_pass(1, gP1);   // NF is workgroup-uniform
    if (NF > 2) frame_pass(2, gP2);

    // SI-log term and conversion depth -> upsampled disparity
    const int grp = b / (cfg.B / cfg.groups);
    const float n_valid = a.stats[8 + 4 * grp], m1 = a.stats[9 + 4 * grp], var = a.stats[10 + 4 * grp];
    const float k_si = a.beam ? a.g[1] / (float)cfg.groups * 0.1f / (sqrtf(var) * n_valid) : 0.f;
    const float k_l1 = a.beam ? a.g[1] / (float)cfg.groups * 0.001f / n_valid * cfg.si_depth_scale : 0.f;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int y = y0t + ty * PPT + i;
        if (y >= H || x >= W) continue;
        const long p = (long)y * W + x;
        const float sdisp = cm.lo + cm.span * disp_up_at(disp_b, cm, y, x);
        const float depth = frcp(sdisp);
        float dd = d_depth[i];
        if (a.beam) {
            const float d26 = depth * cfg.si_depth_scale;
            const float bd = a.beam[b * P + p] * cfg.si_beam_scale;
            if (cfg.si_mode == 1) {
                if (bd > cfg.si_lo && d26 < 80.f && d26 > cfg.si_lo) dd += d26 > bd ? k_l1 : (d26 < bd ? -k_l1 : 0.f);
            } else if (bd > cfg.si_lo && d26 < 80.f && d26 > cfg.si_lo && fabsf(d26 - bd) < cfg.si_threshold) {
                const float d = logf(d26) - logf(bd);
                dd += k_si * (d - cfg.si_var * m1) * sdisp;
            }
        }
        a.d_up[b * P + p] = -dd * depth * depth * cm.span;
    }
    float gPa[36];
#pragma unroll
    for (int i = 0; i < 12; ++i) { gPa[i] = gP0[i]; gPa[12 + i] = gP1[i]; gPa[24 + i] = gP2[i]; }
    const float s = fd_block_sum_n<36, NWV>(gPa, s_red);
    if (tid < 36) {
        const long blk = ((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        a.part[blk * 36 + tid] = s;
    }
}

// gP[b][f][12] = sum over the tiles of image b.  One workgroup per image: thread (g, k) sums every 7th tile for entry k,
// then the 7 group partials are added in a fixed order (deterministic).
__global__ void __launch_bounds__(256) k_photo_bwd_fin(const float* __restrict__ part, float* __restrict__ gP, int tiles, int NF) {
    __shared__ float red[7][36];
    const int b = blockIdx.x, t = threadIdx.x;
    const int k = t % 36, gidx = t / 36;
    if (gidx < 7) {
        float s = 0.f;
        for (int i = gidx; i < tiles; i += 7) s += part[((long)b * tiles + i) * 36 + k];
        red[gidx][k] = s;
    }
    __syncthreads();
    if (t < NF * 12) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) s += red[j][t];
        gP[(long)b * NF * 12 + t] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone maps (layers.py SSIM module, trainer.py:476-488) — same tile machinery, inputs from HBM.
template <bool SSIM>
__global__ void __launch_bounds__(NT) k_reproj_map(const float* __restrict__ pred, const float* __restrict__ target,
                                                   float* __restrict__ out, long out_bs, int H, int W) {
    __shared__ float s_x[3 * H1 * W1];
    __shared__ float s_y[3 * H1 * W1];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0t = blockIdx.x * TW, y0t = blockIdx.y * TH, b = blockIdx.z;
    const long P = (long)H * W;
    for (int i = tid; i < H1 * W1; i += NT) {
        const int hy = i / W1, hx = i - hy * W1;
        const long o = (long)refl_clamp(y0t - 1 + hy, H) * W + refl_clamp(x0t - 1 + hx, W);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            s_x[c * H1 * W1 + i] = pred[((long)b * 3 + c) * P + o];
            s_y[c * H1 * W1 + i] = target[((long)b * 3 + c) * P + o];
        }
    }
    __syncthreads();
    float L[PPT];
    column_losses<W1, 1, SSIM>(s_x, s_y, H1 * W1, tx, ty, L);
    const int x = x0t + tx;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int y = y0t + ty * PPT + i;
        if (y < H && x < W) out[b * out_bs + (long)y * W + x] = L[i];
    }
}

// single-plane SSIM loss map
__global__ void __launch_bounds__(NT) k_ssim_fwd(const float* __restrict__ xg, const float* __restrict__ yg,
                                                 float* __restrict__ out, int H, int W) {
    __shared__ float s_x[H1 * W1];
    __shared__ float s_y[H1 * W1];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0t = blockIdx.x * TW, y0t = blockIdx.y * TH;
    const long base = (long)blockIdx.z * H * W;
    for (int i = tid; i < H1 * W1; i += NT) {
        const int hy = i / W1, hx = i - hy * W1;
        const long o = base + (long)refl_clamp(y0t - 1 + hy, H) * W + refl_clamp(x0t - 1 + hx, W);
        s_x[i] = xg[o]; s_y[i] = yg[o];
    }
    __syncthreads();
    const int x = x0t + tx;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int y = y0t + ty * PPT + i;
        if (y >= H || x >= W) continue;
        float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Syy = 0.f, Sxy = 0.f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float xv = s_x[(ty * PPT + i + dy) * W1 + tx + dx], yv = s_y[(ty * PPT + i + dy) * W1 + tx + dx];
                Sx += xv; Sy += yv; Sxx += xv * xv; Syy += yv * yv; Sxy += xv * yv;
            }
        out[base + (long)y * W + x] = ssim_from_sums(Sx, Sy, Sxx, Syy, Sxy);
    }
}

// gradient of sum(ssim(x,y)*g) w.r.t. x, single plane (SSIM is symmetric: call with (y,x) for d/dy)
__global__ void __launch_bounds__(NT) k_ssim_bwd(const float* __restrict__ xg, const float* __restrict__ yg,
                                                 const float* __restrict__ gg, float* __restrict__ gx, int H, int W) {
    __shared__ float s_x[H2 * W2];
    __shared__ float s_y[H2 * W2];
    __shared__ float s_coef[3 * H1 * W1];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int x0t = blockIdx.x * TW, y0t = blockIdx.y * TH;
    const long base = (long)blockIdx.z * H * W;
    for (int i = tid; i < H2 * W2; i += NT) {
        const int hy = i / W2, hx = i - hy * W2;
        const long o = base + (long)refl_clamp(y0t - 2 + hy, H) * W + refl_clamp(x0t - 2 + hx, W);
        s_x[i] = xg[o]; s_y[i] = yg[o];
    }
    __syncthreads();
    for (int i = tid; i < H1 * W1; i += NT) {
        const int hy = i / W1, hx = i - hy * W1;
        const int py = y0t - 1 + hy, px = x0t - 1 + hx;
        float ca = 0.f, cb = 0.f, cc = 0.f;
        if (py >= 0 && py < H && px >= 0 && px < W) {
            float Sx = 0.f, Sy = 0.f, Sxx = 0.f, Syy = 0.f, Sxy = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float xv = s_x[(hy + dy) * W2 + hx + dx], yv = s_y[(hy + dy) * W2 + hx + dx];
                    Sx += xv; Sy += yv; Sxx += xv * xv; Syy += yv * yv; Sxy += xv * yv;
                }
            ssim_coefs(Sx, Sy, Sxx, Syy, Sxy, gg[base + (long)py * W + px], ca, cb, cc);
        }
        s_coef[i] = ca; s_coef[H1 * W1 + i] = cb; s_coef[2 * H1 * W1 + i] = cc;
    }
    __syncthreads();
    const int x = x0t + tx;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
        const int y = y0t + ty * PPT + i;
        if (y >= H || x >= W) continue;
        const int o = (ty * PPT + i + 2) * W2 + tx + 2;
        const float sa = fold_sum(s_coef, y, x, y0t, x0t, H, W);
        const float sb = fold_sum(s_coef + H1 * W1, y, x, y0t, x0t, H, W);
        const float sc = fold_sum(s_coef + 2 * H1 * W1, y, x, y0t, x0t, H, W);
        gx[base + (long)y * W + x] = div9(sa + 2.f * s_x[o] * sb + s_y[o] * sc);
    }
}

inline dim3 tile_grid(int B, int H, int W) { return dim3(fd_cdiv(W, TW), fd_cdiv(H, TH), B); }
inline long tile_count(int B, int H, int W) { return (long)B * fd_cdiv(W, TW) * fd_cdiv(H, TH); }

int check_cfg(const fd_photo_cfg* c, const char* who) {
    FD_REQUIRE(c, "%s: cfg is NULL", who);
    FD_REQUIRE(c->B > 0 && c->H >= 4 && c->W >= 4 && c->Hs > 0 && c->Ws > 0 && c->Hs <= c->H && c->Ws <= c->W,
               "%s: bad sizes B=%d H=%d W=%d Hs=%d Ws=%d", who, c->B, c->H, c->W, c->Hs, c->Ws);
    FD_REQUIRE(c->NF >= 1 && c->NF <= 3, "%s: NF must be 1, 2 or 3 (got %d)", who, c->NF);
    FD_REQUIRE(c->groups >= 1 && c->groups <= 16 && c->B % c->groups == 0, "%s: batch %d not divisible into %d groups", who, c->B,
               c->groups);
    FD_REQUIRE(c->min_depth > 0 && c->max_depth > c->min_depth, "%s: bad depth range", who);
    FD_REQUIRE(c->si_mode == 0 || c->si_mode == 1, "%s: si_mode must be 0 (SI-log) or 1 (masked L1), got %d", who, c->si_mode);
    return 0;
}

}  // namespace

extern "C" long fd_photo_ws_floats(int B, int H, int W) { return tile_count(B, H, W) * 4; }
extern "C" long fd_photo_bwd_ws_floats(int B, int H, int W) { return tile_count(B, H, W) * 36 + (long)B * H * W; }

extern "C" int fd_photo_fwd_ex(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                               const float* const* src, const float* target, const float* ident, const float* noise,
                               const float* beam, const float* mask, uint8_t* sel, float* depth_out, float* sample_out,
                               float* color_out, float* reproj_out, float* ws, float* out, void* stream) {
    if (int rc = check_cfg(cfg, "fd_photo_fwd")) return rc;
    FD_REQUIRE(disp && inv_K && P && src && target && sel && ws && out, "fd_photo_fwd: NULL argument");
    FD_REQUIRE(!(noise && !ident), "fd_photo_fwd: noise without ident");
    PhotoArgs a;
    a.cfg = *cfg;
    a.disp = disp; a.inv_K = inv_K; a.P = P;
    a.src[0] = src[0]; a.src[1] = cfg->NF > 1 ? src[1] : src[0]; a.src[2] = cfg->NF > 2 ? src[2] : src[0];
    FD_REQUIRE(a.src[0] && a.src[1] && a.src[2], "fd_photo_fwd: NULL source image");
    FD_REQUIRE(!(mask && ident), "fd_photo_fwd: the predictive mask replaces automasking (trainer.py:117-119); pass one of them");
    a.target = target; a.ident = ident; a.noise = noise; a.beam = beam; a.mask = mask;
    a.sel = sel; a.depth_out = depth_out; a.sample_out = sample_out; a.color_out = color_out; a.reproj_out = reproj_out; a.ws = ws;
    dim3 grid = tile_grid(cfg->B, cfg->H, cfg->W);
    hipStream_t st = (hipStream_t)stream;
    if (cfg->use_ssim) hipLaunchKernelGGL(k_photo_fwd<true>, grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(k_photo_fwd<false>, grid, dim3(NT), 0, st, a);
    FD_LAUNCH_CHECK("fd_photo_fwd");
    const float count = (float)cfg->B * (float)cfg->H * (float)cfg->W;
    hipLaunchKernelGGL(k_photo_finalize, dim3(cfg->groups), dim3(256), 0, st, ws,
                       (int)(tile_count(cfg->B, cfg->H, cfg->W) / cfg->groups), cfg->groups, count, cfg->si_var, beam ? 1 : 0, cfg->si_mode, out);
    FD_LAUNCH_CHECK("fd_photo_finalize");
    hipLaunchKernelGGL(k_photo_finalize2, dim3(1), dim3(64), 0, st, out, cfg->groups, count);
    FD_LAUNCH_CHECK("fd_photo_finalize2");
    return 0;
}

extern "C" int fd_photo_fwd(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                            const float* const* src, const float* target, const float* ident, const float* noise,
                            const float* beam, uint8_t* sel, float* depth_out, float* sample_out, float* color_out,
                            float* ws, float* out, void* stream) {
    return fd_photo_fwd_ex(cfg, disp, inv_K, P, src, target, ident, noise, beam, nullptr, sel, depth_out, sample_out, color_out,
                           nullptr, ws, out, stream);
}

// defined in geometry.hip
extern "C" int fd_bilinear_up_bwd(const float*, float*, int, int, int, int, int, void*);

extern "C" int fd_photo_bwd_ex(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                               const float* const* src, const float* target, const float* beam, const float* mask,
                               const uint8_t* sel, int has_ident, const float* stats, const float* g, float* d_disp, float* gP,
                               float* ws, void* stream) {
    if (int rc = check_cfg(cfg, "fd_photo_bwd")) return rc;
    FD_REQUIRE(disp && inv_K && P && src && target && sel && stats && g && d_disp && gP && ws,
               "fd_photo_bwd: NULL argument");
    PhotoBwdArgs a;
    a.cfg = *cfg;
    a.disp = disp; a.inv_K = inv_K; a.P = P;
    a.src[0] = src[0]; a.src[1] = cfg->NF > 1 ? src[1] : src[0]; a.src[2] = cfg->NF > 2 ? src[2] : src[0];
    FD_REQUIRE(a.src[0] && a.src[1] && a.src[2], "fd_photo_bwd: NULL source image");
    a.target = target; a.beam = beam; a.mask = mask; a.sel = sel; a.stats = stats; a.g = g;
    const long ntile = tile_count(cfg->B, cfg->H, cfg->W);
    a.part = ws;
    const bool same = cfg->Hs == cfg->H && cfg->Ws == cfg->W;
    a.d_up = same ? d_disp : ws + ntile * 36;
    a.has_ident = has_ident;
    dim3 grid = tile_grid(cfg->B, cfg->H, cfg->W);
    hipStream_t st = (hipStream_t)stream;
    if (a.cfg.use_ssim) hipLaunchKernelGGL(k_photo_bwd<true>, grid, dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(k_photo_bwd<false>, grid, dim3(NT), 0, st, a);
    FD_LAUNCH_CHECK("fd_photo_bwd");
    hipLaunchKernelGGL(k_photo_bwd_fin, dim3(cfg->B), dim3(256), 0, st, ws, gP, (int)(ntile / cfg->B), cfg->NF);
    FD_LAUNCH_CHECK("fd_photo_bwd_fin");
    if (!same) return fd_bilinear_up_bwd(a.d_up, d_disp, cfg->B, cfg->Hs, cfg->Ws, cfg->H, cfg->W, stream);
    return 0;
}

extern "C" int fd_photo_bwd(const fd_photo_cfg* cfg, const float* disp, const float* inv_K, const float* P,
                            const float* const* src, const float* target, const float* beam, const uint8_t* sel,
                            int has_ident, const float* stats, const float* g, float* d_disp, float* gP, float* ws,
                            void* stream) {
    return fd_photo_bwd_ex(cfg, disp, inv_K, P, src, target, beam, nullptr, sel, has_ident, stats, g, d_disp, gP, ws, stream);
}

extern "C" int fd_reproj_loss_map(const float* pred, const float* target, float* out, long out_batch_stride, int B,
                                  int H, int W, int use_ssim, void* stream) {
    FD_REQUIRE(pred && target && out && B > 0 && H >= 4 && W >= 4 && out_batch_stride >= (long)H * W,
               "fd_reproj_loss_map: bad args");
    dim3 grid = tile_grid(B, H, W);
    if (use_ssim)
        hipLaunchKernelGGL(k_reproj_map<true>, grid, dim3(NT), 0, (hipStream_t)stream, pred, target, out,
                           out_batch_stride, H, W);
    else
        hipLaunchKernelGGL(k_reproj_map<false>, grid, dim3(NT), 0, (hipStream_t)stream, pred, target, out,
                           out_batch_stride, H, W);
    FD_LAUNCH_CHECK("fd_reproj_loss_map");
    return 0;
}

extern "C" int fd_ssim_fwd(const float* x, const float* y, float* out, int B, int C, int H, int W, void* stream) {
    FD_REQUIRE(x && y && out && B > 0 && C > 0 && H >= 4 && W >= 4, "fd_ssim_fwd: bad args");
    hipLaunchKernelGGL(k_ssim_fwd, tile_grid(B * C, H, W), dim3(NT), 0, (hipStream_t)stream, x, y, out, H, W);
    FD_LAUNCH_CHECK("fd_ssim_fwd");
    return 0;
}

extern "C" int fd_ssim_bwd(const float* x, const float* y, const float* g, float* gx, float* gy, int B, int C, int H,
                           int W, void* stream) {
    FD_REQUIRE(x && y && g && (gx || gy) && B > 0 && C > 0 && H >= 4 && W >= 4, "fd_ssim_bwd: bad args");
    if (gx) {
        hipLaunchKernelGGL(k_ssim_bwd, tile_grid(B * C, H, W), dim3(NT), 0, (hipStream_t)stream, x, y, g, gx, H, W);
        FD_LAUNCH_CHECK("fd_ssim_bwd(x)");
    }
    if (gy) {
        hipLaunchKernelGGL(k_ssim_bwd, tile_grid(B * C, H, W), dim3(NT), 0, (hipStream_t)stream, y, x, g, gy, H, W);
        FD_LAUNCH_CHECK("fd_ssim_bwd(y)");
    }
    return 0;
}
