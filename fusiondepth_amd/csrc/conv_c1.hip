// 3x3 stride-1 pad-1 convolutions with ONE output channel: the decoder's dispconv(s) (depth_decoder.py:59-61: Conv3x3(num_ch_dec[s], 1)
// + sigmoid at four scales; refiner / completor heads) - forward and data gradient as stencils.
//
// As a GEMM these layers have M = 1: the implicit-GEMM kernel pads them to a 32-row MFMA tile (204 us for 16 -> 1 at 192x640, batch 12:
// 1/32 of the matrix work is real), the data gradient (Cout = 1 is no multiple of 16) went through the generic gather GEMM on the
// padded grid plus a full fold pass (56 + 54 us).  Both are pure streaming problems - C x 9 multiply-adds per pixel against 4 C bytes
// read (forward) or written (data gradient):
//   forward        y[p]    = act(b + sum_c sum_t W[c][t] x[c][pad(p + t - 1)])           one thread per pixel, rows coalesced, the
//                                                                                       neighbours come from L1;
//   data gradient  gx[c][p] = sum_t W[c][t] S_t[p],   S_t[p] = sum over the pre-images pp of p under the padding of gy[pp - (t - 1)]
//                  - with reflect padding pixel row 1 is also the image of row -1, row H-2 of row H (columns alike), so the adjoint of
//                  ReflectionPad2d(1) is folded into the nine tap sums (no padded-grid tensor, no fold pass); S_t is shared by all C
//                  channels: 9 multiply-adds and one coalesced store per channel.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include <stdlib.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ float c1_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v > 0.f ? v : expm1f(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    if (act == 4) return tanhf(v);
    return v;
}

// grid: x = pixel blocks of 256 along a plane, y = image
__global__ void __launch_bounds__(256) k_conv3x3_c1_fwd(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int C, int H, int W, int pad_mode, int act) {
    extern __shared__ float sw[];                         // W[c][t]: 9 C floats
    for (int i = threadIdx.x; i < 9 * C; i += 256) sw[i] = Wt[i];
    __syncthreads();
    const int hw = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int y = p / W, x = p - y * W;
    const bool refl = pad_mode == 1;
    // the nine source offsets inside a plane (or -1: zero padding)
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        bool ok = true;
        if (yy < 0) { ok = refl; yy = 1; } else if (yy >= H) { ok = refl; yy = H - 2; }
        if (xx < 0) { ok = ok && refl; xx = 1; } else if (xx >= W) { ok = ok && refl; xx = W - 2; }
        off[t] = ok ? yy * W + xx : -1;
    }
    const float* xp = X + (size_t)blockIdx.y * C * hw;
    float acc = bias ? bias[0] : 0.f;
    for (int c = 0; c < C; ++c) {
        const float* q = xp + (size_t)c * hw;
        const float* w = sw + 9 * c;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(w[t], off[t] >= 0 ? q[off[t]] : 0.f, acc);
    }
    Y[(size_t)blockIdx.y * hw + p] = c1_act(acc, act);
}

// Small planes with many channels (dispconv(2), dispconv(3): 64 / 128 channels at 48x160 / 24x80): one thread per pixel leaves a few
// dozen workgroups looping over all channels.  Here a workgroup takes 64 pixels and its four waves a quarter of the channels each; the
// four partial sums meet in LDS and are added in wave order (deterministic).
__global__ void __launch_bounds__(256) k_conv3x3_c1_fwd_cs(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                           float* __restrict__ Y, int C, int H, int W, int pad_mode, int act) {
    extern __shared__ float sw[];                         // W[c][t]: 9 C floats, then 4 x 64 partial sums
    float* part = sw + 9 * C;
    for (int i = threadIdx.x; i < 9 * C; i += 256) sw[i] = Wt[i];
    __syncthreads();
    const int hw = H * W;
    const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int p = blockIdx.x * 64 + lane;
    const bool live = p < hw;
    const int pc = live ? p : 0;
    const int y = pc / W, x = pc - y * W;
    const bool refl = pad_mode == 1;
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        bool ok = true;
        if (yy < 0) { ok = refl; yy = 1; } else if (yy >= H) { ok = refl; yy = H - 2; }
        if (xx < 0) { ok = ok && refl; xx = 1; } else if (xx >= W) { ok = ok && refl; xx = W - 2; }
        off[t] = ok ? yy * W + xx : -1;
    }
    const float* xp = X + (size_t)blockIdx.y * C * hw;
    float acc = 0.f;
    const int c_lo = grp * ((C + 3) / 4), c_hi = min(C, c_lo + (C + 3) / 4);
    for (int c = c_lo; c < c_hi; ++c) {
        const float* q = xp + (size_t)c * hw;
        const float* w = sw + 9 * c;
#pragma unroll
        for (int t = 0; t < 9; ++t) acc = fmaf(w[t], off[t] >= 0 ? q[off[t]] : 0.f, acc);
    }
    part[grp * 64 + lane] = acc;
    __syncthreads();
    if (grp == 0 && live) {
        const float v = (bias ? bias[0] : 0.f) + ((part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]));
        Y[(size_t)blockIdx.y * hw + p] = c1_act(v, act);
    }
}

// The same for four consecutive pixels of a row per thread (W % 4 == 0, 16-byte aligned planes): per channel and tap row one 16-byte
// load + the two neighbours instead of twelve single loads - the one-pixel kernel spent its time issuing loads that hit L1
// (68 us for 16 -> 1 at 192x640, batch 12, against ~20 us of HBM traffic).
__global__ void __launch_bounds__(256) k_conv3x3_c1_fwd4(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                         float* __restrict__ Y, int C, int H, int W, int pad_mode, int act) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < 9 * C; i += 256) sw[i] = Wt[i];
    __syncthreads();
    const int W4 = W >> 2, hw = H * W;
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= H * W4) return;
    const int y = q / W4, x = (q - y * W4) * 4;
    const bool refl = pad_mode == 1;
    // row offsets of the three tap rows (-1: zero padding), columns of the left / right neighbour (-1: zero)
    int ro[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int yy = y + a - 1;
        bool ok = true;
        if (yy < 0) { ok = refl; yy = 1; } else if (yy >= H) { ok = refl; yy = H - 2; }
        ro[a] = ok ? yy * W : -1;
    }
    const int xl = x > 0 ? x - 1 : (refl ? 1 : -1), xr = x + 4 < W ? x + 4 : (refl ? W - 2 : -1);
    const float* xp = X + (size_t)blockIdx.y * C * hw;
    const float b0 = bias ? bias[0] : 0.f;
    float acc[4] = {b0, b0, b0, b0};
    for (int c = 0; c < C; ++c) {
        const float* pl = xp + (size_t)c * hw;
        const float* w = sw + 9 * c;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (ro[a] >= 0) {
                const float4 m = *reinterpret_cast<const float4*>(pl + ro[a] + x);
                v[1] = m.x; v[2] = m.y; v[3] = m.z; v[4] = m.w;
                if (xl >= 0) v[0] = pl[ro[a] + xl];
                if (xr >= 0) v[5] = pl[ro[a] + xr];
            }
            const float w0 = w[3 * a], w1 = w[3 * a + 1], w2 = w[3 * a + 2];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(w2, v[i + 2], fmaf(w1, v[i + 1], fmaf(w0, v[i], acc[i])));
        }
    }
    float4 o;
    o.x = c1_act(acc[0], act); o.y = c1_act(acc[1], act); o.z = c1_act(acc[2], act); o.w = c1_act(acc[3], act);
    *reinterpret_cast<float4*>(Y + (size_t)blockIdx.y * hw + y * W + x) = o;
}

// gx[n][c][p] (+= nothing: plain store) from gy[n][0][.]; Wt = the layer's W[0][c][ky][kx]
// XIN != NULL: the layer's input is the OUTPUT of activation in_act and its producer expects the gradient w.r.t. the PRE-activation
__global__ void __launch_bounds__(256) k_conv3x3_c1_dgrad(const float* __restrict__ GY, const float* __restrict__ Wt, float* __restrict__ GX,
                                                          int C, int H, int W, int pad_mode, const float* __restrict__ XIN, int in_act) {
    extern __shared__ float sw[];
    for (int i = threadIdx.x; i < 9 * C; i += 256) sw[i] = Wt[i];
    __syncthreads();
    const int hw = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= hw) return;
    const int y = p / W, x = p - y * W;
    const bool refl = pad_mode == 1;
    // pre-images of (y, x) on the padded grid: itself, and the padding cells that mirror onto it (row -1 -> 1, row H -> H - 2;
    // with H == 3 row 1 has all three)
    int py[3], px[3], ny = 0, nx = 0;
    py[ny++] = y; px[nx++] = x;
    if (refl) {
        if (y == 1) py[ny++] = -1;
        if (y == H - 2) py[ny++] = H;
        if (x == 1) px[nx++] = -1;
        if (x == W - 2) px[nx++] = W;
    }
    const float* g = GY + (size_t)blockIdx.y * hw;
    float S[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) S[t] = 0.f;
    auto add = [&](int yy0, int xx0) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int yy = yy0 - (t / 3 - 1), xx = xx0 - (t % 3 - 1);        // the output pixel whose tap t reads padded cell (yy0, xx0)
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) S[t] += g[yy * W + xx];
        }
    };
    for (int a = 0; a < ny; ++a)
        for (int b = 0; b < nx; ++b) add(py[a], px[b]);
    float* o = GX + (size_t)blockIdx.y * C * hw + p;
    const float* xi = XIN ? XIN + (size_t)blockIdx.y * C * hw + p : nullptr;
    for (int c = 0; c < C; ++c) {
        const float* w = sw + 9 * c;
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) v = fmaf(w[t], S[t], v);
        if (xi) {
            const float a = xi[(size_t)c * hw];
            v *= in_act == 1 ? (a > 0.f ? 1.f : 0.f) : in_act == 2 ? (a > 0.f ? 1.f : a + 1.f) : in_act == 3 ? a * (1.f - a) : 1.f - a * a;
        }
        o[(size_t)c * hw] = v;
    }
}
}  // namespace

bool c1_shape_ok(const fd_conv_desc* d) {
    if (!fd_tun().conv_c1) return false;                   // 0: Cout = 1 layers stay on the GEMM kernels (A/B timing, tests)
    return d->Cout == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->in_norm && d->H >= 2 && d->W >= 2 &&
           d->Cin <= 1024 && (long)d->N * d->Cin * d->H * d->W < (1L << 29);
}

int c1_fwd_launch(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    const size_t lds = sizeof(float) * 9 * d->Cin;
    if (d->Cin >= 32 && (long)d->N * d->H * d->W < 262144) {            // few pixels, many channels: split the channels over the waves
        const dim3 grid((unsigned)fd_cdiv((long)d->H * d->W, 64), (unsigned)d->N);
        hipLaunchKernelGGL(k_conv3x3_c1_fwd_cs, grid, dim3(256), lds + sizeof(float) * 256, st, x, w, bias, y, d->Cin, d->H, d->W, d->pad_mode, d->act);
        FD_LAUNCH_CHECK("k_conv3x3_c1_fwd_cs");
        return 0;
    }
    if (d->W % 4 == 0 && d->W >= 8 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) {      // (planes are multiples of 4 floats then)
        const dim3 grid((unsigned)fd_cdiv((long)d->H * (d->W / 4), 256), (unsigned)d->N);
        hipLaunchKernelGGL(k_conv3x3_c1_fwd4, grid, dim3(256), lds, st, x, w, bias, y, d->Cin, d->H, d->W, d->pad_mode, d->act);
        FD_LAUNCH_CHECK("k_conv3x3_c1_fwd4");
        return 0;
    }
    const dim3 grid((unsigned)fd_cdiv((long)d->H * d->W, 256), (unsigned)d->N);
    hipLaunchKernelGGL(k_conv3x3_c1_fwd, grid, dim3(256), lds, st, x, w, bias, y, d->Cin, d->H, d->W, d->pad_mode, d->act);
    FD_LAUNCH_CHECK("k_conv3x3_c1_fwd");
    return 0;
}

int c1_dgrad_launch(const fd_conv_desc* d, const float* gy, const float* w, float* gx, hipStream_t st, const float* x_in, int in_act) {
    const dim3 grid((unsigned)fd_cdiv((long)d->H * d->W, 256), (unsigned)d->N);
    hipLaunchKernelGGL(k_conv3x3_c1_dgrad, grid, dim3(256), sizeof(float) * 9 * d->Cin, st, gy, w, gx, d->Cin, d->H, d->W, d->pad_mode, x_in, in_act);
    FD_LAUNCH_CHECK("k_conv3x3_c1_dgrad");
    return 0;
}
