// Edge-aware smoothness loss on the mean-normalised disparity.
// Reference: layers.py:235-248 (get_smooth_loss) + trainer.py:569-571 (mean normalisation).
// Small tensors (<= B*H*W floats), HBM-bound; three light kernels forward, three backward.
#include "../../include/fdhip.h"
#include "fd_common.h"

namespace {

constexpr int NT = 256;
constexpr int PIX_PER_BLOCK = NT * 4;

__device__ __forceinline__ float img_grad(const float* __restrict__ img, long P, long p, long q) {
    // mean over the 3 colour channels of |I[p] - I[q]|
    return (fabsf(img[p] - img[q]) + fabsf(img[P + p] - img[P + q]) + fabsf(img[2 * P + p] - img[2 * P + q])) / 3.0f;
}

// one workgroup per image: mean over H*W (reference: mean over H then over W — same value).  Round 5: float4 loads, four in flight
// (the scalar loop - 120 dependent rounds for a 192x640 plane - took 179 us at the head of the smoothness chain, forward AND backward,
// beside a 255 us photometric kernel on the step's serial section).
__global__ void __launch_bounds__(1024) k_image_mean(const float* __restrict__ x, float* __restrict__ mean, long P) {
    __shared__ float red[16];
    const float* xb = x + (long)blockIdx.x * P;
    float v[1] = {0.f};
    if ((P & 3) == 0 && (((uintptr_t)xb) & 15) == 0) {
        const float4* q = reinterpret_cast<const float4*>(xb);
        const long n4 = P >> 2;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        long i = threadIdx.x;
        for (; i + 3072 < n4; i += 4096) {
            const float4 u0 = q[i], u1 = q[i + 1024], u2 = q[i + 2048], u3 = q[i + 3072];
            a0 += (u0.x + u0.y) + (u0.z + u0.w); a1 += (u1.x + u1.y) + (u1.z + u1.w);
            a2 += (u2.x + u2.y) + (u2.z + u2.w); a3 += (u3.x + u3.y) + (u3.z + u3.w);
        }
        for (; i < n4; i += 1024) { const float4 u = q[i]; a0 += (u.x + u.y) + (u.z + u.w); }
        v[0] = (a0 + a1) + (a2 + a3);
    } else {
        for (long i = threadIdx.x; i < P; i += 1024) v[0] += xb[i];
    }
    const float s = fd_block_sum_n<1, 16>(v, red);
    if (threadIdx.x == 0) mean[blockIdx.x] = s / (float)P;
}

// partial sums of the x- and y-edge terms
__global__ void __launch_bounds__(NT) k_smooth_fwd(const float* __restrict__ disp, const float* __restrict__ img,
                                                   const float* __restrict__ mean, float* __restrict__ part, int H,
                                                   int W) {
    __shared__ float red[4 * 2];
    const int b = blockIdx.y;
    const long P = (long)H * W;
    const float* d = disp + b * P;
    const float* im = img + (long)b * 3 * P;
    const float inv = mean ? 1.0f / (mean[b] + 1e-7f) : 1.0f;
    float acc[2] = {0.f, 0.f};
    for (long p = (long)blockIdx.x * PIX_PER_BLOCK + threadIdx.x, k = 0; k < 4 && p < P; p += NT, ++k) {
        const int x = (int)(p % W), y = (int)(p / W);
        const float c = d[p] * inv;
        if (x < W - 1) acc[0] += fabsf(c - d[p + 1] * inv) * expf(-img_grad(im, P, p, p + 1));
        if (y < H - 1) acc[1] += fabsf(c - d[p + W] * inv) * expf(-img_grad(im, P, p, p + W));
    }
    const float s = fd_block_sum_n<2, 4>(acc, red);
    if (threadIdx.x < 2) part[((long)b * gridDim.x + blockIdx.x) * 2 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(NT) k_smooth_fin(const float* __restrict__ part, int n, float nx, float ny,
                                                   float* __restrict__ out) {
    __shared__ float red[4 * 2];
    float acc[2] = {0.f, 0.f};
    for (int i = threadIdx.x; i < n; i += NT) { acc[0] += part[2 * i]; acc[1] += part[2 * i + 1]; }
    const float s = fd_block_sum_n<2, 4>(acc, red);
    __shared__ float tot[2];
    if (threadIdx.x < 2) tot[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = tot[0] / nx + tot[1] / ny;
}

__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }

// gradient w.r.t. the normalised disparity (gather over the <=4 incident edges), written already
// divided by (mean+eps); per-block partial of sum_p g_n[p]*disp[p] for the mean-normalisation term
__global__ void __launch_bounds__(NT) k_smooth_bwd(const float* __restrict__ disp, const float* __restrict__ img,
                                                   const float* __restrict__ mean, const float* __restrict__ g,
                                                   float* __restrict__ d_disp, float* __restrict__ part, int H, int W,
                                                   float nx, float ny) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const long P = (long)H * W;
    const float* d = disp + b * P;
    const float* im = img + (long)b * 3 * P;
    const float inv = mean ? 1.0f / (mean[b] + 1e-7f) : 1.0f;
    const float gx = g[0] / nx, gy = g[0] / ny;
    float acc[1] = {0.f};
    for (long p = (long)blockIdx.x * PIX_PER_BLOCK + threadIdx.x, k = 0; k < 4 && p < P; p += NT, ++k) {
        const int x = (int)(p % W), y = (int)(p / W);
        const float c = d[p] * inv;
        float gn = 0.f;
        if (x < W - 1) gn += gx * sgn(c - d[p + 1] * inv) * expf(-img_grad(im, P, p, p + 1));
        if (x > 0) gn -= gx * sgn(d[p - 1] * inv - c) * expf(-img_grad(im, P, p - 1, p));
        if (y < H - 1) gn += gy * sgn(c - d[p + W] * inv) * expf(-img_grad(im, P, p, p + W));
        if (y > 0) gn -= gy * sgn(d[p - W] * inv - c) * expf(-img_grad(im, P, p - W, p));
        d_disp[b * P + p] = gn * inv;
        acc[0] += gn * d[p];
    }
    const float s = fd_block_sum_n<1, 4>(acc, red);
    if (threadIdx.x == 0) part[(long)b * gridDim.x + blockIdx.x] = s;
}

// corr[b] = (sum_p g_n d) / ((mean+eps)^2 * P)
__global__ void k_smooth_corr(const float* __restrict__ part, const float* __restrict__ mean, float* __restrict__ corr,
                              int nblk, long P) {
    const int b = blockIdx.x;
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < nblk; ++i) s += part[(long)b * nblk + i];
    const float m = mean[b] + 1e-7f;
    corr[b] = s / (m * m * (float)P);
}

__global__ void k_sub_per_image(float* __restrict__ x, const float* __restrict__ corr, long P) {
    const int b = blockIdx.y;
    const float c = corr[b];
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x)
        x[b * P + p] -= c;
}

inline int nblk(int H, int W) { return fd_cdiv((long)H * W, PIX_PER_BLOCK); }

}  // namespace

// ws layout: [B] mean | [B] corr | [B*nblk*2] partials
extern "C" long fd_smooth_ws_floats(int B, int H, int W) { return 2L * B + 2L * B * nblk(H, W); }

extern "C" int fd_smooth_fwd(const float* disp, const float* img, float* out, float* ws, int B, int H, int W,
                             int normalize, void* stream) {
    FD_REQUIRE(disp && img && out && ws && B > 0 && H > 1 && W > 1, "fd_smooth_fwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)H * W;
    float* mean = ws;
    float* part = ws + 2 * B;
    if (normalize) {
        hipLaunchKernelGGL(k_image_mean, dim3(B), dim3(1024), 0, st, disp, mean, P);
        FD_LAUNCH_CHECK("fd_smooth_fwd(mean)");
    }
    const int nb = nblk(H, W);
    hipLaunchKernelGGL(k_smooth_fwd, dim3(nb, B), dim3(NT), 0, st, disp, img, normalize ? mean : nullptr, part, H, W);
    FD_LAUNCH_CHECK("fd_smooth_fwd");
    hipLaunchKernelGGL(k_smooth_fin, dim3(1), dim3(NT), 0, st, part, nb * B, (float)B * (float)H * (float)(W - 1),
                       (float)B * (float)(H - 1) * (float)W, out);
    FD_LAUNCH_CHECK("fd_smooth_fwd(fin)");
    return 0;
}

extern "C" int fd_smooth_bwd(const float* disp, const float* img, const float* g, float* d_disp, float* ws, int B, int H,
                             int W, int normalize, void* stream) {
    FD_REQUIRE(disp && img && g && d_disp && ws && B > 0 && H > 1 && W > 1, "fd_smooth_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    const long P = (long)H * W;
    float* mean = ws;
    float* corr = ws + B;
    float* part = ws + 2 * B;
    if (normalize) {
        hipLaunchKernelGGL(k_image_mean, dim3(B), dim3(1024), 0, st, disp, mean, P);
        FD_LAUNCH_CHECK("fd_smooth_bwd(mean)");
    }
    const int nb = nblk(H, W);
    hipLaunchKernelGGL(k_smooth_bwd, dim3(nb, B), dim3(NT), 0, st, disp, img, normalize ? mean : nullptr, g, d_disp, part,
                       H, W, (float)B * (float)H * (float)(W - 1), (float)B * (float)(H - 1) * (float)W);
    FD_LAUNCH_CHECK("fd_smooth_bwd");
    if (normalize) {
        hipLaunchKernelGGL(k_smooth_corr, dim3(B), dim3(64), 0, st, part, mean, corr, nb, P);
        FD_LAUNCH_CHECK("fd_smooth_bwd(corr)");
        int gx = fd_cdiv(P, 256);
        gx = gx > 1024 ? 1024 : gx;
        hipLaunchKernelGGL(k_sub_per_image, dim3(gx, B), dim3(256), 0, st, d_disp, corr, P);
        FD_LAUNCH_CHECK("fd_smooth_bwd(sub)");
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// trainer.py:569-596: loss_s = photo_s + w * smooth_s / 2^s;  total = (sum_s loss_s + sum_s si_s) / n_scales.
// The reference does this with ~10 scalar tensor ops per scale (and as many in the backward pass); here one tiny kernel
// each way, on the loss path's serial section.
struct CombineArgs { const float* photo[4]; const float* smooth[4]; const float* si[4]; };
namespace {
__global__ void k_combine_losses(CombineArgs a, int n, float w, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float total = 0.f;
    for (int s = 0; s < n; ++s) {                       // same operation order as the reference's Python loop
        const float loss = a.photo[s][0] + w * a.smooth[s][0] / (float)(1 << s);
        total = total + loss;
        out[s] = loss;
        if (a.si[s]) total = total + a.si[s][0];
    }
    out[n] = total / (float)n;
}
__global__ void k_combine_losses_bwd(const float* __restrict__ g, int n, float w, float* __restrict__ grads) {
    const int s = threadIdx.x;
    if (blockIdx.x != 0 || s >= n) return;
    const float gt = g[0] / (float)n;
    grads[s] = gt;                                      // d total / d photo_s
    grads[n + s] = gt * w / (float)(1 << s);            // d total / d smooth_s
    grads[2 * n + s] = gt;                              // d total / d si_s
}
}  // namespace

extern "C" int fd_combine_losses_fwd(const float* const* photo, const float* const* smooth, const float* const* si, int n_scales,
                                     float smooth_weight, float* out, void* stream) {
    FD_REQUIRE(photo && smooth && si && out && n_scales >= 1 && n_scales <= 4, "fd_combine_losses_fwd: bad args");
    CombineArgs a = {};
    for (int s = 0; s < n_scales; ++s) {
        FD_REQUIRE(photo[s] && smooth[s], "fd_combine_losses_fwd: NULL loss term at scale %d", s);
        a.photo[s] = photo[s]; a.smooth[s] = smooth[s]; a.si[s] = si[s];
    }
    hipLaunchKernelGGL(k_combine_losses, dim3(1), dim3(64), 0, (hipStream_t)stream, a, n_scales, smooth_weight, out);
    FD_LAUNCH_CHECK("fd_combine_losses_fwd");
    return 0;
}
extern "C" int fd_combine_losses_bwd(const float* g_total, int n_scales, float smooth_weight, float* grads, void* stream) {
    FD_REQUIRE(g_total && grads && n_scales >= 1 && n_scales <= 4, "fd_combine_losses_bwd: bad args");
    hipLaunchKernelGGL(k_combine_losses_bwd, dim3(1), dim3(64), 0, (hipStream_t)stream, g_total, n_scales, smooth_weight, grads);
    FD_LAUNCH_CHECK("fd_combine_losses_bwd");
    return 0;
}

