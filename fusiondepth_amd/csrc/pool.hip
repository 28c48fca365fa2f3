// Memory-bound glue kernels of the network stack: 3x3/s2 max-pool (resnet_encoder.py:98), decoder input
// assembly = nearest x2 upsample + skip concat + feature fusion (depth_decoder.py:69-83, layers.py:229-232),
// activation backward, axpby, spatial mean (pose_decoder.py:44-46), depth metrics (layers.py:284-302) and
// the Adam update (trainer.py:129,247).  All 1 thread per output element, coalesced along W.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include <stdint.h>

namespace {

constexpr int NT = 256;
inline int ew_blocks(long n) {
    long b = (n + NT - 1) / NT;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (n); i += (long)gridDim.x * NT)
// Plane-indexed launches: blockIdx.y strides over the (n, c) planes, blockIdx.x over the elements of one plane, so the
// per-element index math is one 32-bit division instead of three 64-bit ones.
#define PLANE_LOOP(pl, r, planes, psize)                                  \
    for (long pl = blockIdx.y; pl < (planes); pl += gridDim.y)            \
        for (int r = blockIdx.x * NT + threadIdx.x; r < (psize); r += gridDim.x * NT)
inline dim3 plane_grid(long planes, long psize) {
    long bx = (psize + NT - 1) / NT;
    if (bx > 64) bx = 64;
    return dim3((unsigned)(bx < 1 ? 1 : bx), (unsigned)(planes < 1 ? 1 : (planes > 32768 ? 32768 : planes)));
}

// ---- max-pool 3x3 stride 2 pad 1; first maximum in (kh,kw) raster order wins, as ATen does (val > max) ----
__global__ void __launch_bounds__(NT) k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y,
                                                    uint8_t* __restrict__ idx, long planes, int H, int W, int Ho, int Wo) {
    PLANE_LOOP(pl, r, planes, Ho * Wo) {
        const int ho = r / Wo, wo = r - ho * Wo;
        const float* p = x + pl * H * W;
        float best = 0.f;
        int bi = -1;
        // branch-free: every tap is loaded from a clamped address; taps outside the image never win
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            const bool vh = (unsigned)h < (unsigned)H;
            const int hc = h < 0 ? 0 : (h >= H ? H - 1 : h);
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int w = wo * 2 - 1 + kw;
                const bool ok = vh & ((unsigned)w < (unsigned)W);
                const int wc = w < 0 ? 0 : (w >= W ? W - 1 : w);
                const float v = p[hc * W + wc];
                const bool take = ok & ((bi < 0) | (v > best) | (v != v));
                best = take ? v : best;
                bi = take ? kh * 3 + kw : bi;
            }
        }
        y[pl * Ho * Wo + r] = best;
        idx[pl * Ho * Wo + r] = (uint8_t)bi;
    }
}
// Gather form of the scatter-add.  One thread owns a 2x2 block of input pixels (rows 2p, 2p+1; columns 2q, 2q+1); the only
// pooling windows that can select them are (p, q), (p, q+1), (p+1, q), (p+1, q+1), so 4 index bytes + 4 gradients give the
// 4 outputs with no branches and contiguous accesses across the lanes.
__global__ void __launch_bounds__(NT) k_maxpool_bwd(const float* __restrict__ gy, const uint8_t* __restrict__ idx,
                                                    float* __restrict__ gx, long planes, int H, int W, int Ho, int Wo) {
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    PLANE_LOOP(pl, r, planes, Hb * Wb) {
        const int p = r / Wb, q = r - p * Wb;
        const float* g = gy + pl * Ho * Wo;
        const uint8_t* ix = idx + pl * Ho * Wo;
        const int p1 = p + 1 < Ho ? p + 1 : p, q1 = q + 1 < Wo ? q + 1 : q;      // clamped; masked below
        const bool vp = p + 1 < Ho, vq = q + 1 < Wo;
        const int i00 = ix[p * Wo + q], i01 = vq ? ix[p * Wo + q1] : 255, i10 = vp ? ix[p1 * Wo + q] : 255,
                  i11 = (vp & vq) ? ix[p1 * Wo + q1] : 255;
        const float g00 = g[p * Wo + q], g01 = g[p * Wo + q1], g10 = g[p1 * Wo + q], g11 = g[p1 * Wo + q1];
        // window (p,q): taps (kh,kw) in {1,2}x{1,2} hit this block; (p,q+1): kw = 0, kh in {1,2}; (p+1,q): kh = 0, kw in {1,2};
        // (p+1,q+1): tap 0.  Sums are taken in window raster order (p,q), (p,q+1), (p+1,q), (p+1,q+1).
        const float o00 = (i00 == 4 ? g00 : 0.f);
        const float o01 = (i00 == 5 ? g00 : 0.f) + (i01 == 3 ? g01 : 0.f);
        const float o10 = (i00 == 7 ? g00 : 0.f) + (i10 == 1 ? g10 : 0.f);
        const float o11 = (((i00 == 8 ? g00 : 0.f) + (i01 == 6 ? g01 : 0.f)) + (i10 == 2 ? g10 : 0.f)) + (i11 == 0 ? g11 : 0.f);
        float* o = gx + pl * H * W + (2 * p) * W + 2 * q;
        const bool h1 = 2 * p + 1 < H, w1 = 2 * q + 1 < W;
        o[0] = o00;
        if (w1) o[1] = o01;
        if (h1) { o[W] = o10; if (w1) o[W + 1] = o11; }
    }
}

// ---- decoder input assembly ----
__global__ void __launch_bounds__(NT) k_upcat_fwd(const float* __restrict__ a, const float* __restrict__ s1,
                                                  const float* __restrict__ s2, const float* __restrict__ s3,
                                                  float* __restrict__ out, int N, int Ca, int Cs, int C3, int h, int w) {
    const int H = 2 * h, W = 2 * w, Ct = Ca + Cs + C3;
    PLANE_LOOP(pl, r, (long)N * Ct, H * W) {
        const int y = r / W, x = r - y * W;
        const long b = pl / Ct;
        const int c = (int)(pl - b * Ct);
        const long i = pl * H * W + r;
        float v;
        if (c < Ca) v = a[((b * Ca + c) * h + (y >> 1)) * w + (x >> 1)];
        else if (c < Ca + Cs) {
            const long o = ((b * Cs + (c - Ca)) * H + y) * W + x;
            v = s1[o];
            if (s2) v += s2[o];
        } else v = s3[((b * C3 + (c - Ca - Cs)) * H + y) * W + x];
        out[i] = v;
    }
}
// Four output pixels per thread (w even, 16-byte aligned tensors): the up-sampled part reads two source pixels and writes one 16-byte
// row piece, the skip part moves 16-byte pieces - the one-pixel kernels above ran at half the HBM rate on the decoder's serial chain
// (upsample + concatenation 0.26 ms, channel split 0.10 ms, up-sample adjoint 0.05 ms per training step).
__global__ void __launch_bounds__(NT) k_upcat_fwd4(const float* __restrict__ a, const float* __restrict__ s1, const float* __restrict__ s2,
                                                   const float* __restrict__ s3, float* __restrict__ out, int N, int Ca, int Cs, int C3,
                                                   int h, int w) {
    const int H = 2 * h, W = 2 * w, W4 = W >> 2, Ct = Ca + Cs + C3;
    PLANE_LOOP(pl, r, (long)N * Ct, H * W4) {
        const int y = r / W4, x = (r - y * W4) * 4;
        const long b = pl / Ct;
        const int c = (int)(pl - b * Ct);
        float4 v;
        if (c < Ca) {
            const float2 q = *reinterpret_cast<const float2*>(a + ((b * Ca + c) * h + (y >> 1)) * w + (x >> 1));
            v = make_float4(q.x, q.x, q.y, q.y);
        } else if (c < Ca + Cs) {
            const long o = ((b * Cs + (c - Ca)) * H + y) * W + x;
            v = *reinterpret_cast<const float4*>(s1 + o);
            if (s2) { const float4 u = *reinterpret_cast<const float4*>(s2 + o); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
        } else v = *reinterpret_cast<const float4*>(s3 + ((b * C3 + (c - Ca - Cs)) * H + y) * W + x);
        *reinterpret_cast<float4*>(out + pl * H * W + (long)y * W + x) = v;
    }
}
// two output pixels of ga per thread from two 16-byte pieces of gout
// derivative of the activation from its OUTPUT v (1 ReLU, 2 ELU(alpha=1), 3 sigmoid, 4 tanh), as k_act_bwd
__device__ __forceinline__ float act_deriv(float v, int act) {
    if (act == 1) return v > 0.f ? 1.f : 0.f;
    if (act == 2) return v > 0.f ? 1.f : v + 1.f;
    if (act == 3) return v * (1.f - v);
    return 1.f - v * v;
}
// `a_out` != NULL: `a` is the OUTPUT of an activation whose producer expects the gradient w.r.t. its PRE-activation: ga *= act'(a)
__global__ void __launch_bounds__(NT) k_upcat_bwd_a2(const float* __restrict__ gout, float* __restrict__ ga, int N, int Ca, int Ct, int h, int w,
                                                     const float* __restrict__ a_out, int a_act) {
    const int H = 2 * h, W = 2 * w, w2 = w >> 1;
    PLANE_LOOP(pl, r, (long)N * Ca, h * w2) {
        const int y = r / w2, x = (r - y * w2) * 2;
        const long b = pl / Ca;
        const int c = (int)(pl - b * Ca);
        const float* g = gout + ((b * Ct + c) * H + 2 * y) * W + 2 * x;
        const float4 t = *reinterpret_cast<const float4*>(g), u = *reinterpret_cast<const float4*>(g + W);
        float2 o;
        o.x = (t.x + t.y) + (u.x + u.y); o.y = (t.z + t.w) + (u.z + u.w);
        if (a_out) {
            const float2 av = *reinterpret_cast<const float2*>(a_out + pl * h * w + (long)y * w + x);
            o.x *= act_deriv(av.x, a_act); o.y *= act_deriv(av.y, a_act);
        }
        *reinterpret_cast<float2*>(ga + pl * h * w + (long)y * w + x) = o;
    }
}
__global__ void __launch_bounds__(NT) k_slice_channels4(const float* __restrict__ src, float* __restrict__ dst, int N, int Ct, int c0, int Cn,
                                                        long plane4) {
    PLANE_LOOP(pl, r, (long)N * Cn, plane4) {
        const long b = pl / Cn;
        const int c = (int)(pl - b * Cn);
        reinterpret_cast<float4*>(dst)[pl * plane4 + r] = reinterpret_cast<const float4*>(src)[(b * Ct + c0 + c) * plane4 + r];
    }
}
__global__ void __launch_bounds__(NT) k_upcat_bwd_a(const float* __restrict__ gout, float* __restrict__ ga, int N, int Ca,
                                                    int Ct, int h, int w, const float* __restrict__ a_out, int a_act) {
    const int H = 2 * h, W = 2 * w;
    PLANE_LOOP(pl, r, (long)N * Ca, h * w) {
        const int y = r / w, x = r - y * w;
        const long b = pl / Ca;
        const int c = (int)(pl - b * Ca);
        const long i = pl * h * w + r;
        const float* g = gout + ((b * Ct + c) * H + 2 * y) * W + 2 * x;
        float v = (g[0] + g[1]) + (g[W] + g[W + 1]);
        if (a_out) v *= act_deriv(a_out[i], a_act);
        ga[i] = v;
    }
}
__global__ void __launch_bounds__(NT) k_slice_channels(const float* __restrict__ src, float* __restrict__ dst, int N,
                                                       int Ct, int c0, int Cn, long plane) {
    PLANE_LOOP(pl, r, (long)N * Cn, plane) {
        const long b = pl / Cn;
        const int c = (int)(pl - b * Cn);
        dst[pl * plane + r] = src[(b * Ct + c0 + c) * plane + r];
    }
}
__global__ void __launch_bounds__(NT) k_up2_fwd(const float* __restrict__ x, float* __restrict__ y, long planes, int h,
                                                int w) {
    const int H = 2 * h, W = 2 * w;
    PLANE_LOOP(pl, r, planes, H * W) {
        const int yy = r / W, xx = r - yy * W;
        y[pl * H * W + r] = x[(pl * h + (yy >> 1)) * w + (xx >> 1)];
    }
}
__global__ void __launch_bounds__(NT) k_up2_bwd(const float* __restrict__ gy, float* __restrict__ gx, long planes, int h,
                                                int w) {
    const int W = 2 * w;
    PLANE_LOOP(pl, r, planes, h * w) {
        const int yy = r / w, xx = r - yy * w;
        const float* g = gy + (pl * 2 * h + 2 * yy) * W + 2 * xx;
        gx[pl * h * w + r] = (g[0] + g[1]) + (g[W] + g[W + 1]);
    }
}

// ---- elementwise ----
__global__ void __launch_bounds__(NT) k_act_bwd(const float* __restrict__ y, const float* __restrict__ gy,
                                                float* __restrict__ out, long n, int act) {
    GRID_STRIDE(i, n) {
        const float v = y[i], g = gy[i];
        float d;
        if (act == 1) d = v > 0.f ? 1.f : 0.f;
        else if (act == 2) d = v > 0.f ? 1.f : v + 1.f;        // ELU(alpha=1): y = e^x - 1  =>  dy/dx = y + 1
        else if (act == 3) d = v * (1.f - v);
        else if (act == 4) d = 1.f - v * v;
        else d = 1.f;
        out[i] = g * d;
    }
}
__global__ void __launch_bounds__(NT) k_axpby(const float* __restrict__ a, const float* __restrict__ b,
                                              float* __restrict__ out, long n, float alpha, float beta) {
    GRID_STRIDE(i, n) out[i] = alpha * a[i] + beta * b[i];
}

// networks/resnet_encoder.py:94  x = (input_image - 0.45) / 0.225 (a true division, as the reference rounds it)
__global__ void __launch_bounds__(NT) k_input_normalize(const float* __restrict__ x, float* __restrict__ y, long n,
                                                        float mean, float std) {
    GRID_STRIDE(i, n) y[i] = (x[i] - mean) / std;
}

// The pose encoders' input (trainer.py:336-351: torch.cat([color_aug[f_i], color_aug[f_j]], 1) per source frame, then resnet_encoder.py:94)
// assembled and normalised in ONE pass: piece p = `imgs` whole images (C * HW floats each, contiguous in the source) that go to
// images dst_img[p] .. of the stacked tensor at channel offset dst_ch[p]; out has `Ct` channels per image.
struct StackArgs { const float* src[16]; int dst_img[16]; int dst_ch[16]; int n_pieces, imgs, C, Ct; long HW; float mean, std; int normalize; };
__global__ void __launch_bounds__(NT) k_stack_normalize(StackArgs a, float* __restrict__ out) {
    const int piece = blockIdx.y / a.imgs, b = blockIdx.y - piece * a.imgs;
    const long n4 = ((long)a.C * a.HW) >> 2;                          // HW % 4 == 0 (checked by the launcher)
    const float4* s = reinterpret_cast<const float4*>(a.src[piece] + (long)b * a.C * a.HW);
    float4* d = reinterpret_cast<float4*>(out + ((long)(a.dst_img[piece] + b) * a.Ct + a.dst_ch[piece]) * a.HW);
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < n4; i += (long)gridDim.x * NT) {
        float4 v = s[i];
        if (a.normalize) { v.x = (v.x - a.mean) / a.std; v.y = (v.y - a.mean) / a.std; v.z = (v.z - a.mean) / a.std; v.w = (v.w - a.mean) / a.std; }
        d[i] = v;
    }
}

// one wave per plane: out = scale * mean(plane)
__global__ void __launch_bounds__(64) k_spatial_mean(const float* __restrict__ x, float* __restrict__ out, long plane_size,
                                                     float scale) {
    const float* p = x + (long)blockIdx.x * plane_size;
    float s = 0.f;
    for (long i = threadIdx.x; i < plane_size; i += 64) s += p[i];
    s = fd_wave_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = scale * (s / (float)plane_size);
}
__global__ void __launch_bounds__(NT) k_spatial_mean_bwd(const float* __restrict__ gout, float* __restrict__ gx,
                                                         long planes, long plane_size, float scale) {
    const long n = planes * plane_size;
    GRID_STRIDE(i, n) gx[i] = gout[i / plane_size] * (scale / (float)plane_size);
}

// ---- depth metrics ----
__global__ void __launch_bounds__(NT) k_depth_err_part(const float* __restrict__ gt, const float* __restrict__ pr, long n,
                                                       float* __restrict__ part) {
    __shared__ float red[4 * 7];
    float a[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    GRID_STRIDE(i, n) {
        const float g = gt[i], p = pr[i];
        const float d = g - p, th = fmaxf(g / p, p / g), l = logf(g) - logf(p);
        a[0] += fabsf(d) / g; a[1] += d * d / g; a[2] += d * d; a[3] += l * l;
        a[4] += th < 1.25f ? 1.f : 0.f; a[5] += th < 1.25f * 1.25f ? 1.f : 0.f; a[6] += th < 1.25f * 1.25f * 1.25f ? 1.f : 0.f;
    }
    const float s = fd_block_sum_n<7, 4>(a, red);
    if (threadIdx.x < 7) part[(long)blockIdx.x * 7 + threadIdx.x] = s;
}
__global__ void k_depth_err_fin(const float* __restrict__ part, int nblk, float n, float* __restrict__ out) {
    const int t = threadIdx.x;
    if (t >= 7) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part[(long)i * 7 + t];
    s /= n;
    out[t] = (t == 2 || t == 3) ? sqrtf(s) : s;
}

// ---- Adam ----
__global__ void __launch_bounds__(NT) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, long n, float step_size, float b1, float b2, float eps,
                                             float sqrt_bc2, float gscale) {
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
    }
}

__global__ void k_adam_tick(float* __restrict__ state) { state[0] += 1.0f; }
__global__ void __launch_bounds__(NT) k_adam_dev(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, long n, const float* __restrict__ state, float b1,
                                                 float b2, float eps, float gscale) {
    const float step = state[0], lr = state[1];
    const float bc1 = 1.f - powf(b1, step), bc2 = 1.f - powf(b2, step);
    const float step_size = lr / bc1, sqrt_bc2 = sqrtf(bc2);
    GRID_STRIDE(i, n) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * (mi / (sqrtf(vi) / sqrt_bc2 + eps));
    }
}

}  // namespace

extern "C" int fd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int N, int C, int H, int W, void* stream) {
    FD_REQUIRE(x && y && idx && N > 0 && C > 0 && H > 0 && W > 0, "fd_maxpool3x3s2_fwd: bad args");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long planes = (long)N * C;
    hipLaunchKernelGGL(k_maxpool_fwd, plane_grid(planes, (long)Ho * Wo), dim3(NT), 0, (hipStream_t)stream, x, y, idx,
                       planes, H, W, Ho, Wo);
    FD_LAUNCH_CHECK("fd_maxpool3x3s2_fwd");
    return 0;
}
extern "C" int fd_maxpool3x3s2_bwd(const float* gy, const uint8_t* idx, float* gx, int N, int C, int H, int W,
                                   void* stream) {
    FD_REQUIRE(gy && idx && gx && N > 0 && C > 0 && H > 0 && W > 0, "fd_maxpool3x3s2_bwd: bad args");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long planes = (long)N * C;
    hipLaunchKernelGGL(k_maxpool_bwd, plane_grid(planes, (long)((H + 1) / 2) * ((W + 1) / 2)), dim3(NT), 0, (hipStream_t)stream, gy, idx, gx,
                       planes, H, W, Ho, Wo);
    FD_LAUNCH_CHECK("fd_maxpool3x3s2_bwd");
    return 0;
}

extern "C" int fd_upcat_fwd(const float* a, const float* s1, const float* s2, const float* s3, float* out, int N, int Ca,
                            int Cs, int C3, int h, int w, void* stream) {
    FD_REQUIRE(a && out && N > 0 && Ca > 0 && Cs >= 0 && C3 >= 0 && h > 0 && w > 0, "fd_upcat_fwd: bad args");
    FD_REQUIRE((Cs == 0 || s1) && (C3 == 0 || s3) && !(s2 && !s1), "fd_upcat_fwd: missing skip tensor");
    const long n = (long)N * (Ca + Cs + C3) * 4 * h * w;
    const uintptr_t al = (uintptr_t)a | (uintptr_t)s1 | (uintptr_t)s2 | (uintptr_t)s3 | (uintptr_t)out;
    if (w % 2 == 0 && (al & 15) == 0)            // planes of 4 h w and h w floats: multiples of 4 resp. 2 floats - every row piece stays aligned
        hipLaunchKernelGGL(k_upcat_fwd4, plane_grid((long)N * (Ca + Cs + C3), (long)h * w), dim3(NT), 0, (hipStream_t)stream, a, s1, s2, s3, out,
                           N, Ca, Cs, C3, h, w);
    else
    hipLaunchKernelGGL(k_upcat_fwd, plane_grid((long)N * (Ca + Cs + C3), 4L * h * w), dim3(NT), 0, (hipStream_t)stream, a, s1, s2, s3, out, N, Ca, Cs,
                       C3, h, w);
    FD_LAUNCH_CHECK("fd_upcat_fwd");
    return 0;
}
static int upcat_bwd_impl(const float* gout, float* ga, float* gs, float* g3, int N, int Ca, int Cs, int C3, int h, int w, const float* a_out,
                          int a_act, void* stream) {
    FD_REQUIRE(gout && N > 0 && Ca > 0 && h > 0 && w > 0, "fd_upcat_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const int Ct = Ca + Cs + C3;
    const long plane = 4L * h * w;
    const bool vec = w % 2 == 0 && (((uintptr_t)gout | (uintptr_t)ga | (uintptr_t)gs | (uintptr_t)g3) & 15) == 0;
    if (ga) {
        const bool veca = vec && (!a_out || ((uintptr_t)a_out & 7) == 0);
        if (veca) hipLaunchKernelGGL(k_upcat_bwd_a2, plane_grid((long)N * Ca, (long)h * (w / 2)), dim3(NT), 0, st, gout, ga, N, Ca, Ct, h, w, a_out, a_act);
        else hipLaunchKernelGGL(k_upcat_bwd_a, plane_grid((long)N * Ca, (long)h * w), dim3(NT), 0, st, gout, ga, N, Ca, Ct, h, w, a_out, a_act);
        FD_LAUNCH_CHECK("fd_upcat_bwd(a)");
    }
    if (gs && Cs > 0) {
        if (vec) hipLaunchKernelGGL(k_slice_channels4, plane_grid((long)N * Cs, plane / 4), dim3(NT), 0, st, gout, gs, N, Ct, Ca, Cs, plane / 4);
        else hipLaunchKernelGGL(k_slice_channels, plane_grid((long)N * Cs, plane), dim3(NT), 0, st, gout, gs, N, Ct, Ca, Cs, plane);
        FD_LAUNCH_CHECK("fd_upcat_bwd(s)");
    }
    if (g3 && C3 > 0) {
        if (vec) hipLaunchKernelGGL(k_slice_channels4, plane_grid((long)N * C3, plane / 4), dim3(NT), 0, st, gout, g3, N, Ct, Ca + Cs, C3, plane / 4);
        else hipLaunchKernelGGL(k_slice_channels, plane_grid((long)N * C3, plane), dim3(NT), 0, st, gout, g3, N, Ct, Ca + Cs, C3, plane);
        FD_LAUNCH_CHECK("fd_upcat_bwd(3)");
    }
    return 0;
}
extern "C" int fd_upcat_bwd(const float* gout, float* ga, float* gs, float* g3, int N, int Ca, int Cs, int C3, int h,
                            int w, void* stream) {
    return upcat_bwd_impl(gout, ga, gs, g3, N, Ca, Cs, C3, h, w, nullptr, 0, stream);
}
extern "C" int fd_upcat_bwd_act(const float* gout, const float* a_out, int a_act, float* ga, float* gs, float* g3, int N, int Ca, int Cs,
                                int C3, int h, int w, void* stream) {
    FD_REQUIRE(a_out && ga && a_act >= 1 && a_act <= 4, "fd_upcat_bwd_act: needs the activation output, ga and an activation id 1..4");
    return upcat_bwd_impl(gout, ga, gs, g3, N, Ca, Cs, C3, h, w, a_out, a_act, stream);
}
extern "C" int fd_upsample2x_fwd(const float* x, float* y, long planes, int h, int w, void* stream) {
    FD_REQUIRE(x && y && planes > 0 && h > 0 && w > 0, "fd_upsample2x_fwd: bad args");
    hipLaunchKernelGGL(k_up2_fwd, plane_grid(planes, 4L * h * w), dim3(NT), 0, (hipStream_t)stream, x, y, planes, h, w);
    FD_LAUNCH_CHECK("fd_upsample2x_fwd");
    return 0;
}
extern "C" int fd_upsample2x_bwd(const float* gy, float* gx, long planes, int h, int w, void* stream) {
    FD_REQUIRE(gy && gx && planes > 0 && h > 0 && w > 0, "fd_upsample2x_bwd: bad args");
    hipLaunchKernelGGL(k_up2_bwd, plane_grid(planes, (long)h * w), dim3(NT), 0, (hipStream_t)stream, gy, gx, planes, h, w);
    FD_LAUNCH_CHECK("fd_upsample2x_bwd");
    return 0;
}

extern "C" int fd_act_bwd(const float* y, const float* gy, float* gpre, long n, int act, void* stream) {
    FD_REQUIRE(y && gy && gpre && n >= 0 && act >= 0 && act <= 4, "fd_act_bwd: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_act_bwd, dim3(ew_blocks(n)), dim3(NT), 0, (hipStream_t)stream, y, gy, gpre, n, act);
    FD_LAUNCH_CHECK("fd_act_bwd");
    return 0;
}
extern "C" int fd_axpby(const float* a, const float* b, float* out, long n, float alpha, float beta, void* stream) {
    FD_REQUIRE(a && b && out && n >= 0, "fd_axpby: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_axpby, dim3(ew_blocks(n)), dim3(NT), 0, (hipStream_t)stream, a, b, out, n, alpha, beta);
    FD_LAUNCH_CHECK("fd_axpby");
    return 0;
}
extern "C" int fd_input_normalize(const float* x, float* y, long n, float mean, float std, void* stream) {
    FD_REQUIRE(x && y && n >= 0 && std != 0.f, "fd_input_normalize: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_input_normalize, dim3(ew_blocks(n)), dim3(NT), 0, (hipStream_t)stream, x, y, n, mean, std);
    FD_LAUNCH_CHECK("fd_input_normalize");
    return 0;
}
extern "C" int fd_stack_normalize(const float* const* src, const int* dst_img, const int* dst_ch, int n_pieces, int imgs, int C, int Ct,
                                  int H, int W, float* out, int normalize, float mean, float std, void* stream) {
    FD_REQUIRE(src && dst_img && dst_ch && out && n_pieces > 0 && n_pieces <= 16 && imgs > 0 && C > 0 && Ct >= C && H > 0 && W > 0 && std != 0.f,
               "fd_stack_normalize: bad args (at most 16 pieces)");
    FD_REQUIRE(((long)H * W) % 4 == 0 && ((uintptr_t)out & 15) == 0, "fd_stack_normalize: planes must be multiples of 4 floats, 16-byte aligned");
    StackArgs a;
    for (int p = 0; p < 16; ++p) { a.src[p] = nullptr; a.dst_img[p] = 0; a.dst_ch[p] = 0; }
    for (int p = 0; p < n_pieces; ++p) {
        FD_REQUIRE(src[p] && ((uintptr_t)src[p] & 15) == 0 && dst_img[p] >= 0 && dst_ch[p] >= 0 && dst_ch[p] + C <= Ct, "fd_stack_normalize: bad piece %d", p);
        a.src[p] = src[p]; a.dst_img[p] = dst_img[p]; a.dst_ch[p] = dst_ch[p];
    }
    a.n_pieces = n_pieces; a.imgs = imgs; a.C = C; a.Ct = Ct; a.HW = (long)H * W; a.mean = mean; a.std = std; a.normalize = normalize;
    long bx = ((long)C * H * W / 4 + 4 * NT - 1) / (4 * NT);
    bx = bx < 1 ? 1 : (bx > 256 ? 256 : bx);
    hipLaunchKernelGGL(k_stack_normalize, dim3((unsigned)bx, (unsigned)(n_pieces * imgs)), dim3(NT), 0, (hipStream_t)stream, a, out);
    FD_LAUNCH_CHECK("fd_stack_normalize");
    return 0;
}
extern "C" int fd_spatial_mean_fwd(const float* x, float* out, long planes, long plane_size, float scale, void* stream) {
    FD_REQUIRE(x && out && planes > 0 && plane_size > 0, "fd_spatial_mean_fwd: bad args");
    hipLaunchKernelGGL(k_spatial_mean, dim3((unsigned)planes), dim3(64), 0, (hipStream_t)stream, x, out, plane_size, scale);
    FD_LAUNCH_CHECK("fd_spatial_mean_fwd");
    return 0;
}
extern "C" int fd_spatial_mean_bwd(const float* gout, float* gx, long planes, long plane_size, float scale, void* stream) {
    FD_REQUIRE(gout && gx && planes > 0 && plane_size > 0, "fd_spatial_mean_bwd: bad args");
    hipLaunchKernelGGL(k_spatial_mean_bwd, dim3(ew_blocks(planes * plane_size)), dim3(NT), 0, (hipStream_t)stream, gout, gx,
                       planes, plane_size, scale);
    FD_LAUNCH_CHECK("fd_spatial_mean_bwd");
    return 0;
}
extern "C" int fd_depth_errors(const float* gt, const float* pred, long n, float* out, float* ws, void* stream) {
    FD_REQUIRE(gt && pred && out && ws && n > 0, "fd_depth_errors: bad args");
    int nb = ew_blocks(n);
    nb = nb > 256 ? 256 : nb;
    hipLaunchKernelGGL(k_depth_err_part, dim3(nb), dim3(NT), 0, (hipStream_t)stream, gt, pred, n, ws);
    FD_LAUNCH_CHECK("fd_depth_errors");
    hipLaunchKernelGGL(k_depth_err_fin, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, (float)n, out);
    FD_LAUNCH_CHECK("fd_depth_errors(fin)");
    return 0;
}
extern "C" int fd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float lr,
                            float beta1, float beta2, float eps, float bias_corr1, float bias_corr2, float grad_scale,
                            void* stream) {
    FD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && bias_corr1 > 0 && bias_corr2 > 0, "fd_adam_step: bad args");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_adam, dim3(ew_blocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       lr / bias_corr1, beta1, beta2, eps, sqrtf(bias_corr2), grad_scale);
    FD_LAUNCH_CHECK("fd_adam_step");
    return 0;
}

extern "C" int fd_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float* state,
                                float beta1, float beta2, float eps, float grad_scale, void* stream) {
    FD_REQUIRE(param && grad && exp_avg && exp_avg_sq && state && n >= 0, "fd_adam_step_dev: bad args");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, (hipStream_t)stream, state);
    FD_LAUNCH_CHECK("fd_adam_step_dev(tick)");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_adam_dev, dim3(ew_blocks(n)), dim3(NT), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n,
                       state, beta1, beta2, eps, grad_scale);
    FD_LAUNCH_CHECK("fd_adam_step_dev");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// evaluate_depth.py:62-70 batch_post_process_disparity: blend of the disparity of an image and of its mirrored twin
// (Monodepth-v1 post-processing).  float32 disparities, float64 ramp masks and result, exactly the reference's
// numpy expression (np.linspace(0, 1, w): i * (1 / (w - 1)), last element 1; no fused multiply-adds).
namespace {
#pragma clang fp contract(off)
__global__ void k_post_process_disparity(const float* __restrict__ l_disp, const float* __restrict__ r_disp, double* __restrict__ out,
                                         long planes, int H, int W) {
    const double step = 1.0 / (double)(W - 1);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < planes * H * W; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const int xr = W - 1 - x;
        const double lx = (x == W - 1) ? 1.0 : (double)x * step;
        const double lr = (xr == W - 1) ? 1.0 : (double)xr * step;
        double a = 20.0 * (lx - 0.05), b = 20.0 * (lr - 0.05);
        a = a < 0.0 ? 0.0 : (a > 1.0 ? 1.0 : a);
        b = b < 0.0 ? 0.0 : (b > 1.0 ? 1.0 : b);
        const double l_mask = 1.0 - a, r_mask = 1.0 - b;
        const float l = l_disp[i], r = r_disp[i];
        const float m = 0.5f * (l + r);
        out[i] = (r_mask * (double)l + l_mask * (double)r) + ((1.0 - l_mask) - r_mask) * (double)m;
    }
}
}  // namespace

extern "C" int fd_post_process_disparity(const float* l_disp, const float* r_disp, double* out, long planes, int H, int W, void* stream) {
    FD_REQUIRE(l_disp && r_disp && out && planes > 0 && H > 0 && W > 1, "fd_post_process_disparity: bad args");
    hipLaunchKernelGGL(k_post_process_disparity, dim3(fd_cdiv(planes * H * W, 256) > 4096 ? 4096 : fd_cdiv(planes * H * W, 256)),
                       dim3(256), 0, (hipStream_t)stream, l_disp, r_disp, out, planes, H, W);
    FD_LAUNCH_CHECK("fd_post_process_disparity");
    return 0;
}
