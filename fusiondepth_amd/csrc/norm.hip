// BatchNorm2d (training-mode batch statistics) fused with the residual add + ReLU tail of the ResNet blocks.
// Reference: torchvision ResNet BasicBlock/Bottleneck as driven by networks/resnet_encoder.py:95-101 in
// train mode (trainer.py:207-211).  HBM-bound: forward = 1 statistics pass + 1 apply pass, backward = 1
// reduction pass + 1 apply pass; statistics use shifted single-pass sums (large planes: shift = median of 9
// samples of the channel, stored with the partial sums; small planes: first element, with an exact second pass over the
// registers when it turns out to be far from the mean; conv-epilogue partials: Chan-merged (sum, M2)) so var =
// E[(x-k)^2] - E[x-k]^2 does not cancel catastrophically.  Deterministic: per-(channel, slice) partials are combined in
// slice order by every consumer.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include <stdint.h>
#include <stdlib.h>

namespace {

constexpr int NT = 256;

// number of plane slices per (group, channel): enough workgroups to fill the chip, >= 1024 floats of a plane per slice
inline int bn_splits(int N, int C, long HW, int groups) {
    long s = 2048 / ((long)C * groups);
    if (s < 1) s = 1;
    const long max_s = (HW + 1023) / 1024;
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    return (int)(s < 1 ? 1 : s);
}
inline bool bn_vec_ok(long HW, const void* a, const void* b, const void* c, const void* d, const void* e) {
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    return (HW & 3) == 0 && al(a) && al(b) && al(c) && al(d) && al(e);
}

// A statistics / reduction workgroup covers slice s of the plane of channel c for every sample of group g: element
// (n, r) with r in [lo, hi).  VEC: planes are multiples of 4 floats and 16-byte aligned, so each lane moves float4.
template <bool VEC, typename F>
__device__ __forceinline__ void for_channel_slice(int N, long HW, int C, int c, long n0, int s, int splits, F&& f) {
    if (VEC) {
        const long q = HW >> 2, per = (q + splits - 1) / splits;
        const long lo = (long)s * per, hi = lo + per < q ? lo + per : q;
        for (int n = 0; n < N; ++n) {
            const long base = ((n0 + n) * C + c) * HW;
            for (long i = lo + threadIdx.x; i < hi; i += NT) f(base + 4 * i);
        }
    } else {
        const long per = (HW + splits - 1) / splits;
        const long lo = (long)s * per, hi = lo + per < HW ? lo + per : HW;
        for (int n = 0; n < N; ++n) {
            const long base = ((n0 + n) * C + c) * HW;
            for (long i = lo + threadIdx.x; i < hi; i += NT) f(base + i);
        }
    }
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Shift of the single-pass statistics of one (group, channel): the MEDIAN of 9 samples spread over the group's first plane.
// var = E[d^2] - E[d]^2 (d = x - shift) loses eps * (mean - shift)^2 / var: the shift has to sit within a few standard
// deviations of the mean on ANY data.  The first element of the plane (rounds 1-2) is a corner pixel - after the zero-padded
// 7x7 stem over the beam encoder's sparse LiDAR image it was 11 sigma off, variance wrong by 8e-5, the feature by 2.4e-4
// against 1.5e-5 for float32 two-pass arithmetic; a sample MEAN is dragged as far by one outlier.  |mean - median| <= sigma
// for every distribution, and the median of 9 stays near the median.  Sorting network: 19 exchanges (Paeth / Devillard).
__device__ __forceinline__ float bn_shift(const float* __restrict__ plane, long HW) {
    float p[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) p[j] = plane[((2 * j + 1) * HW) / 18];
#define FD_CE(a, b) { const float lo = fminf(p[a], p[b]); p[b] = fmaxf(p[a], p[b]); p[a] = lo; }
    FD_CE(1, 2) FD_CE(4, 5) FD_CE(7, 8) FD_CE(0, 1) FD_CE(3, 4) FD_CE(6, 7) FD_CE(1, 2) FD_CE(4, 5) FD_CE(7, 8) FD_CE(0, 3)
    FD_CE(5, 8) FD_CE(4, 7) FD_CE(3, 6) FD_CE(1, 4) FD_CE(2, 5) FD_CE(4, 7) FD_CE(4, 2) FD_CE(6, 4) FD_CE(4, 2)
#undef FD_CE
    return p[4];
}

// N = samples PER GROUP; group g = blockIdx.z covers samples [g*N, (g+1)*N): statistics are per (group, channel), which
// is exactly what G separate forward passes over the sub-batches would compute (the pose encoders see frames -1 and +1
// as two passes in the reference; here they are one launch with G = 2).
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_stats(const float* __restrict__ x, float* __restrict__ part, float* __restrict__ shifts,
                                                 int N, int C, long HW, int splits) {
    __shared__ float red[4 * 2];
    const int c = blockIdx.x, s = blockIdx.y, g = blockIdx.z;
    const long n0 = (long)g * N;
    const float shift = bn_shift(x + (n0 * C + c) * HW, HW);
    float acc[2] = {0.f, 0.f};
    for_channel_slice<VEC>(N, HW, C, c, n0, s, splits, [&](long o) {
        if (VEC) {
            const float4 v = ld4(x + o);
            const float d0 = v.x - shift, d1 = v.y - shift, d2 = v.z - shift, d3 = v.w - shift;
            acc[0] += (d0 + d1) + (d2 + d3);
            acc[1] += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        } else {
            const float d = x[o] - shift; acc[0] += d; acc[1] += d * d;
        }
    });
    const float r = fd_block_sum_n<2, 4>(acc, red);
    if (threadIdx.x < 2) part[(((long)g * C + c) * splits + s) * 2 + threadIdx.x] = r;
    // the apply pass reads the shift instead of re-deriving it (nine scattered loads + the sorting network) in every workgroup
    if (threadIdx.x == 0 && s == 0) shifts[(long)g * C + c] = shift;
}

struct BnStat { float mean, var; };
__device__ __forceinline__ BnStat bn_finalize(const float* __restrict__ part, const float* __restrict__ shifts, int c, int C,
                                              int g, int splits, float M) {
    float s1 = 0.f, s2 = 0.f;
    const long pb = ((long)g * C + c) * splits;
    for (int s = 0; s < splits; ++s) { s1 += part[(pb + s) * 2]; s2 += part[(pb + s) * 2 + 1]; }
    const float shift = shifts[(long)g * C + c];
    const float m = s1 / M;
    BnStat st;
    st.mean = shift + m;
    st.var = fmaxf(s2 / M - m * m, 0.f);
    return st;
}

// y = relu?( (x-mean)*invstd*w + b + residual? ); grid (plane chunks, N*C)
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_apply_train(const float* __restrict__ x, const float* __restrict__ weight,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       float* __restrict__ y, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, const float* __restrict__ part,
                                                       const float* __restrict__ shifts, int N, int C, long HW, int splits,
                                                       float eps, float momentum, int relu, int G) {
    // N = samples per group; blockIdx.y enumerates (sample, channel) over all G*N samples
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    const float M = (float)N * (float)HW;
    const BnStat st = bn_finalize(part, shifts, c, C, g, splits, M);
    const float invstd = 1.0f / sqrtf(st.var + eps);
    if (blockIdx.x == 0 && n == g * N && threadIdx.x == 0) {   // once per (group, channel)
        save_mean[g * C + c] = st.mean;
        save_invstd[g * C + c] = invstd;
    }
    if (blockIdx.x == 0 && nc < C && threadIdx.x == 0 && running_mean) {
        // running statistics: the G momentum updates are applied in group order by ONE thread, i.e. exactly as G
        // consecutive forward passes would have applied them
        float rm = running_mean[c], rv = running_var[c];
        for (int gg = 0; gg < G; ++gg) {
            const BnStat sg = bn_finalize(part, shifts, c, C, gg, splits, M);
            const float unbiased = M > 1.f ? sg.var * (M / (M - 1.f)) : sg.var;
            rm = (1.f - momentum) * rm + momentum * sg.mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
    const float a = invstd * (weight ? weight[c] : 1.f);
    const float b = (bias ? bias[c] : 0.f) - st.mean * a;
    const long base = (long)nc * HW;
    if (VEC) {
        for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (HW >> 2); i += (long)gridDim.x * NT) {
            float4 v = ld4(x + base + 4 * i);
            v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
            if (residual) { const float4 r = ld4(residual + base + 4 * i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
            st4(y + base + 4 * i, v);
        }
        return;
    }
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        float v = fmaf(x[base + i], a, b);
        if (residual) v += residual[base + i];
        if (relu) v = v > 0.f ? v : 0.f;
        y[base + i] = v;
    }
}

// The same apply pass when the statistics come from the producing convolution's epilogue (fd_conv2d_fwd_stats): `cpart`
// [G*N][C][S][2] = (sum, sum of squares) of the S pixel slots of every (image, channel).  Each workgroup first reduces the N * S
// entries of its (group, channel) - fixed assignment of entries to threads, fixed tree: deterministic - which replaces the
// separate statistics launch (and its pass over x) of the two-launch path.
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_apply_parts(const float* __restrict__ x, const float* __restrict__ weight,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       float* __restrict__ y, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, const float* __restrict__ cpart,
                                                       int N, int C, long HW, int S, float eps, float momentum, int relu, int G) {
    __shared__ float red[4 * 2];
    __shared__ float gst[16][2];
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    const float M = (float)N * (float)HW;
    const bool owner = blockIdx.x == 0 && nc < C && running_mean != nullptr;      // updates the running statistics of channel c
    // partial = (sum, M2 about the partial's own mean) of cnt = HW / S pixels; the group's M2 = sum of M2_i + cnt * (mean_i - mean)^2
    const float cnt = (float)(HW / S), rcnt = 1.0f / cnt;
    auto group_stat = [&](int gg) -> BnStat {
        const int E = N * S;
        BnStat st;
        st.mean = 0.f; st.var = 0.f;
        for (int pass = 0; pass < 2; ++pass) {
            float acc[1] = {0.f};
            for (int e = threadIdx.x; e < E; e += NT) {
                const int ni = e / S, si = e - ni * S;
                const float2 v = *reinterpret_cast<const float2*>(cpart + ((((long)gg * N + ni) * C + c) * S + si) * 2);
                const float dm = v.x * rcnt - st.mean;                     // pass 1 only (st.mean is set by pass 0)
                acc[0] += pass == 0 ? v.x : fmaf(cnt * dm, dm, v.y);
            }
            const float r = fd_block_sum_n<1, 4>(acc, red);
            if (threadIdx.x == 0) gst[gg][pass] = r;
            __syncthreads();
            if (pass == 0) st.mean = gst[gg][0] / M; else st.var = gst[gg][1] / M;
        }
        return st;
    };
    const BnStat st = group_stat(g);
    const float invstd = 1.0f / sqrtf(st.var + eps);
    if (blockIdx.x == 0 && n == g * N && threadIdx.x == 0) {   // once per (group, channel)
        save_mean[g * C + c] = st.mean;
        save_invstd[g * C + c] = invstd;
    }
    if (owner) {                                               // the G momentum updates in group order, as G consecutive passes would
        float rm = 0.f, rv = 0.f;
        if (threadIdx.x == 0) { rm = running_mean[c]; rv = running_var[c]; }
        for (int gg = 0; gg < G; ++gg) {
            const BnStat sg = group_stat(gg);                  // uniform across the workgroup (block reductions inside)
            const float unbiased = M > 1.f ? sg.var * (M / (M - 1.f)) : sg.var;
            rm = (1.f - momentum) * rm + momentum * sg.mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        if (threadIdx.x == 0) { running_mean[c] = rm; running_var[c] = rv; }
    }
    const float a = invstd * (weight ? weight[c] : 1.f);
    const float b = (bias ? bias[c] : 0.f) - st.mean * a;
    const long base = (long)nc * HW;
    if (VEC) {
        for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (HW >> 2); i += (long)gridDim.x * NT) {
            float4 v = ld4(x + base + 4 * i);
            v.x = fmaf(v.x, a, b); v.y = fmaf(v.y, a, b); v.z = fmaf(v.z, a, b); v.w = fmaf(v.w, a, b);
            if (residual) { const float4 r = ld4(residual + base + 4 * i); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            if (relu) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
            st4(y + base + 4 * i, v);
        }
        return;
    }
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        float v = fmaf(x[base + i], a, b);
        if (residual) v += residual[base + i];
        if (relu) v = v > 0.f ? v : 0.f;
        y[base + i] = v;
    }
}

__global__ void __launch_bounds__(NT) k_bn_apply_eval(const float* __restrict__ x, const float* __restrict__ weight,
                                                      const float* __restrict__ bias, const float* __restrict__ residual,
                                                      float* __restrict__ y, const float* __restrict__ running_mean,
                                                      const float* __restrict__ running_var, int C, long HW, float eps,
                                                      int relu) {
    const int nc = blockIdx.y, c = nc % C;
    const float invstd = 1.0f / sqrtf(running_var[c] + eps);
    const float a = invstd * (weight ? weight[c] : 1.f);
    const float b = (bias ? bias[c] : 0.f) - running_mean[c] * a;
    const long base = (long)nc * HW;
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        float v = fmaf(x[base + i], a, b);
        if (residual) v += residual[base + i];
        if (relu) v = v > 0.f ? v : 0.f;
        y[base + i] = v;
    }
}

// partial sums of dy' and dy'*xhat   (dy' = dy masked by the ReLU)
// ``y == nullptr`` with ``relu``: the ReLU mask is recomputed from x (a * x + b > 0 with the forward's per-channel a, b) - a BatchNorm
// without a residual input (bn1 of a BasicBlock, bn1 / bn2 of a Bottleneck) then reads two tensors per pass instead of three.
template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_bwd_reduce(const float* __restrict__ x, const float* __restrict__ y,
                                                      const float* __restrict__ gy, const float* __restrict__ save_mean,
                                                      const float* __restrict__ save_invstd, float* __restrict__ part,
                                                      int N, int C, long HW, int splits, int relu, const float* __restrict__ weight,
                                                      const float* __restrict__ bias) {
    __shared__ float red[4 * 2];
    const int c = blockIdx.x, s = blockIdx.y, g = blockIdx.z;
    const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
    const bool remask = relu && y == nullptr;
    const float ma = invstd * (weight ? weight[c] : 1.f), mb = (bias ? bias[c] : 0.f) - mean * ma;
    float acc[2] = {0.f, 0.f};
    for_channel_slice<VEC>(N, HW, C, c, (long)g * N, s, splits, [&](long o) {
        if (VEC) {
            float4 d = ld4(gy + o);
            const float4 xx = ld4(x + o);
            if (remask) {
                d.x = fmaf(xx.x, ma, mb) > 0.f ? d.x : 0.f; d.y = fmaf(xx.y, ma, mb) > 0.f ? d.y : 0.f;
                d.z = fmaf(xx.z, ma, mb) > 0.f ? d.z : 0.f; d.w = fmaf(xx.w, ma, mb) > 0.f ? d.w : 0.f;
            } else if (relu) {
                const float4 yy = ld4(y + o);
                d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f; d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
            }
            acc[0] += (d.x + d.y) + (d.z + d.w);
            acc[1] += (d.x * ((xx.x - mean) * invstd) + d.y * ((xx.y - mean) * invstd)) +
                      (d.z * ((xx.z - mean) * invstd) + d.w * ((xx.w - mean) * invstd));
        } else {
            float d = gy[o];
            if (remask) { if (!(fmaf(x[o], ma, mb) > 0.f)) d = 0.f; }
            else if (relu && !(y[o] > 0.f)) d = 0.f;
            acc[0] += d;
            acc[1] += d * ((x[o] - mean) * invstd);
        }
    });
    const float r = fd_block_sum_n<2, 4>(acc, red);
    if (threadIdx.x < 2) part[(((long)g * C + c) * splits + s) * 2 + threadIdx.x] = r;
}

template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gy, const float* __restrict__ weight,
                                                     const float* __restrict__ save_mean,
                                                     const float* __restrict__ save_invstd, float* __restrict__ gx,
                                                     float* __restrict__ gweight, float* __restrict__ gbias,
                                                     float* __restrict__ g_res, const float* __restrict__ part, int N,
                                                     int C, long HW, int splits, int relu, int accumulate, int G,
                                                     const float* __restrict__ bias) {
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    float s1 = 0.f, s2 = 0.f;
    const long pb = ((long)g * C + c) * splits;
    for (int s = 0; s < splits; ++s) { s1 += part[(pb + s) * 2]; s2 += part[(pb + s) * 2 + 1]; }
    if (blockIdx.x == 0 && nc < C && threadIdx.x == 0) {   // parameter gradients: sum over the groups, in group order
        float t1 = 0.f, t2 = 0.f;
        for (int gg = 0; gg < G; ++gg) {
            float u1 = 0.f, u2 = 0.f;
            const long qb = ((long)gg * C + c) * splits;
            for (int s = 0; s < splits; ++s) { u1 += part[(qb + s) * 2]; u2 += part[(qb + s) * 2 + 1]; }
            t1 += u1; t2 += u2;
        }
        if (gbias) gbias[c] = (accumulate ? gbias[c] : 0.f) + t1;
        if (gweight) gweight[c] = (accumulate ? gweight[c] : 0.f) + t2;
    }
    const float M = (float)N * (float)HW;
    const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
    const float k = (weight ? weight[c] : 1.f) * invstd;
    const float m1 = s1 / M, m2 = s2 / M;
    const long base = (long)nc * HW;
    const bool remask = relu && y == nullptr;
    const float ma = k, mb = (bias ? bias[c] : 0.f) - mean * ma;
    if (VEC) {
        for (long i = (long)blockIdx.x * NT + threadIdx.x; i < (HW >> 2); i += (long)gridDim.x * NT) {
            const long o = base + 4 * i;
            float4 d = ld4(gy + o);
            const float4 xx = ld4(x + o);
            if (remask) {
                d.x = fmaf(xx.x, ma, mb) > 0.f ? d.x : 0.f; d.y = fmaf(xx.y, ma, mb) > 0.f ? d.y : 0.f;
                d.z = fmaf(xx.z, ma, mb) > 0.f ? d.z : 0.f; d.w = fmaf(xx.w, ma, mb) > 0.f ? d.w : 0.f;
            } else if (relu) {
                const float4 yy = ld4(y + o);
                d.x = yy.x > 0.f ? d.x : 0.f; d.y = yy.y > 0.f ? d.y : 0.f; d.z = yy.z > 0.f ? d.z : 0.f; d.w = yy.w > 0.f ? d.w : 0.f;
            }
            if (g_res) st4(g_res + o, d);
            float4 r;
            r.x = k * (d.x - m1 - ((xx.x - mean) * invstd) * m2);
            r.y = k * (d.y - m1 - ((xx.y - mean) * invstd) * m2);
            r.z = k * (d.z - m1 - ((xx.z - mean) * invstd) * m2);
            r.w = k * (d.w - m1 - ((xx.w - mean) * invstd) * m2);
            st4(gx + o, r);
        }
        return;
    }
    for (long i = (long)blockIdx.x * NT + threadIdx.x; i < HW; i += (long)gridDim.x * NT) {
        float g = gy[base + i];
        if (remask) { if (!(fmaf(x[base + i], ma, mb) > 0.f)) g = 0.f; }
        else if (relu && !(y[base + i] > 0.f)) g = 0.f;
        if (g_res) g_res[base + i] = g;
        const float xh = (x[base + i] - mean) * invstd;
        gx[base + i] = k * (g - m1 - xh * m2);
    }
}

// ---- small planes (ResNet layer3 / layer4: a (group, channel) holds <= 4096 floats): statistics + apply in ONE launch ----------
// One workgroup per channel; the N * HW / 4 float4 of a (group, channel) live in registers (<= SMALL_K per thread), so x is read
// once instead of twice and the two launches of the large-plane path (5-8 us each at these sizes, launch-latency bound) become
// one.  Groups are processed in order by the same workgroup, which makes the in-order running-statistics update and the
// group-ordered parameter-gradient sums local.  Fixed reduction tree: deterministic.
#ifndef FD_BN_SMALL_K
#define FD_BN_SMALL_K 4
#endif
constexpr int SMALL_K = FD_BN_SMALL_K;

__device__ __forceinline__ void block_sum2(float& a, float& b, float* red) {     // result broadcast to every thread
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    a = fd_wave_sum(a); b = fd_wave_sum(b);
    __syncthreads();                                                               // previous readers of `red` are done
    if (lane == 0) { red[wv * 2] = a; red[wv * 2 + 1] = b; }
    __syncthreads();
    a = (red[0] + red[2]) + (red[4] + red[6]);
    b = (red[1] + red[3]) + (red[5] + red[7]);
}

__global__ void __launch_bounds__(NT) k_bn_train_small(const float* __restrict__ x, const float* __restrict__ weight,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       float* __restrict__ y, float* __restrict__ running_mean,
                                                       float* __restrict__ running_var, float* __restrict__ save_mean,
                                                       float* __restrict__ save_invstd, int N, int C, int q, float eps,
                                                       float momentum, int relu, int G) {
    __shared__ float red[8];
    __shared__ float gstat[16][2];
    const int c = blockIdx.x, E = N * q;                 // float4 elements of one (group, channel)
    const float M = (float)N * (float)(4 * q);
    const float wc = weight ? weight[c] : 1.f, bc = bias ? bias[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        // The whole (group, channel) sits in registers.  First try: single-pass sums shifted by the plane's first element (one
        // broadcast load that travels with the data).  If that shift turns out to sit more than 4 standard deviations from the mean
        // (var = E[d^2] - E[d]^2 then loses eps * 16 and more) the deviations are summed again from the registers around the now
        // known mean - exact two-pass statistics, a workgroup-uniform branch taken for the rare channel that needs it.
        const float shift = x[((long)g * N * C + c) * 4 * q];
        float4 v[SMALL_K];
        long off[SMALL_K];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            const int e = threadIdx.x + k * NT;
            off[k] = -1;
            if (e < E) {
                const int n = e / q, i = e - n * q;
                off[k] = (((long)g * N + n) * C + c) * 4 * q + 4 * i;
                v[k] = ld4(x + off[k]);
                const float d0 = v[k].x - shift, d1 = v[k].y - shift, d2 = v[k].z - shift, d3 = v[k].w - shift;
                s1 += (d0 + d1) + (d2 + d3);
                s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
        }
        block_sum2(s1, s2, red);
        const float m = s1 / M;
        const float mean = shift + m;
        float var = fmaxf(s2 / M - m * m, 0.f);
        if (m * m > 16.f * var) {                                   // uniform: s1, s2 are broadcast values
            float t2 = 0.f, dummy = 0.f;
#pragma unroll
            for (int k = 0; k < SMALL_K; ++k) {
                if (off[k] < 0) continue;
                const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
                t2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            block_sum2(t2, dummy, red);
            var = t2 / M;
        }
        const float invstd = 1.0f / sqrtf(var + eps);
        if (threadIdx.x == 0) {
            save_mean[g * C + c] = mean; save_invstd[g * C + c] = invstd;
            gstat[g][0] = mean; gstat[g][1] = var;
        }
        const float a = invstd * wc, b = bc - mean * a;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            if (off[k] < 0) continue;
            float4 r = v[k];
            r.x = fmaf(r.x, a, b); r.y = fmaf(r.y, a, b); r.z = fmaf(r.z, a, b); r.w = fmaf(r.w, a, b);
            if (residual) { const float4 z = ld4(residual + off[k]); r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w; }
            if (relu) { r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f; r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f; }
            st4(y + off[k], r);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && running_mean) {              // the G momentum updates in group order, as G consecutive passes would
        float rm = running_mean[c], rv = running_var[c];
        for (int g = 0; g < G; ++g) {
            const float unbiased = M > 1.f ? gstat[g][1] * (M / (M - 1.f)) : gstat[g][1];
            rm = (1.f - momentum) * rm + momentum * gstat[g][0];
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
}

// k_bn_train_small fed by the SLABS of the F(2x2, 3x3) convolution in front of it (conv_wino.hip: k_conv_wino2d / _m128 write the
// horizontally transformed products S_ri [N][C][H/2][W] per row component ri and channel split): the slab reduction and the vertical
// output transform  y[2 ty] = S0 + S1 + S2,  y[2 ty + 1] = S1 - S2 - S3  of k_wino2d_finish (same order of additions: the convolution
// output is bit-identical) happen where the values are needed - one launch and one pass over y less per deep-layer convolution.
// y (the convolution's output = this BatchNorm's input, needed by both backward passes) is written from here.
__global__ void __launch_bounds__(NT) k_bn_train_small_slabs(const float* __restrict__ slabs, long slab_stride, int ksplit,
                                                             float* __restrict__ xout, const float* __restrict__ weight,
                                                             const float* __restrict__ bias, const float* __restrict__ residual,
                                                             float* __restrict__ y, float* __restrict__ running_mean,
                                                             float* __restrict__ running_var, float* __restrict__ save_mean,
                                                             float* __restrict__ save_invstd, int N, int C, int H, int W, float eps,
                                                             float momentum, int relu, int G) {
    __shared__ float red[8];
    __shared__ float gstat[16][2];
    const int q = (H * W) >> 2, W4 = W >> 2, HT = H >> 1;
    const int c = blockIdx.x, E = N * q;
    const float M = (float)N * (float)(4 * q);
    const float wc = weight ? weight[c] : 1.f, bc = bias ? bias[c] : 0.f;
    for (int g = 0; g < G; ++g) {
        float4 v[SMALL_K];
        long off[SMALL_K];
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            const int e = threadIdx.x + k * NT;
            off[k] = -1;
            if (e < E) {
                const int n = e / q, i = e - n * q;
                const int r = i / W4, col = 4 * (i - r * W4);
                const long pl = ((long)g * N + n) * C + c;
                off[k] = pl * 4 * q + 4 * i;
                const float* sp = slabs + (pl * HT + (r >> 1)) * W + col;
                const int odd = r & 1;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, d = a;           // S_odd, S_odd+1, S_odd+2 summed over the channel splits
                for (int ks = 0; ks < ksplit; ++ks) {
                    const float4 t0 = ld4(sp + (long)(4 * ks + odd) * slab_stride), t1 = ld4(sp + (long)(4 * ks + odd + 1) * slab_stride),
                                 t2 = ld4(sp + (long)(4 * ks + odd + 2) * slab_stride);
                    a.x += t0.x; a.y += t0.y; a.z += t0.z; a.w += t0.w;
                    b.x += t1.x; b.y += t1.y; b.z += t1.z; b.w += t1.w;
                    d.x += t2.x; d.y += t2.y; d.z += t2.z; d.w += t2.w;
                }
                float4 o;
                if (odd) { o.x = (a.x - b.x) - d.x; o.y = (a.y - b.y) - d.y; o.z = (a.z - b.z) - d.z; o.w = (a.w - b.w) - d.w; }
                else { o.x = (a.x + b.x) + d.x; o.y = (a.y + b.y) + d.y; o.z = (a.z + b.z) + d.z; o.w = (a.w + b.w) + d.w; }
                v[k] = o;
                st4(xout + off[k], o);
            }
        }
        // statistics: as k_bn_train_small, with the plane's first element taken from the registers of the thread that holds it
        __shared__ float sh_shift;
        if (threadIdx.x == 0) sh_shift = v[0].x;
        __syncthreads();
        const float shift = sh_shift;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            if (off[k] < 0) continue;
            const float d0 = v[k].x - shift, d1 = v[k].y - shift, d2 = v[k].z - shift, d3 = v[k].w - shift;
            s1 += (d0 + d1) + (d2 + d3);
            s2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        }
        block_sum2(s1, s2, red);
        const float m = s1 / M;
        const float mean = shift + m;
        float var = fmaxf(s2 / M - m * m, 0.f);
        if (m * m > 16.f * var) {
            float t2 = 0.f, dummy = 0.f;
#pragma unroll
            for (int k = 0; k < SMALL_K; ++k) {
                if (off[k] < 0) continue;
                const float d0 = v[k].x - mean, d1 = v[k].y - mean, d2 = v[k].z - mean, d3 = v[k].w - mean;
                t2 += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
            }
            block_sum2(t2, dummy, red);
            var = t2 / M;
        }
        const float invstd = 1.0f / sqrtf(var + eps);
        if (threadIdx.x == 0) {
            save_mean[g * C + c] = mean; save_invstd[g * C + c] = invstd;
            gstat[g][0] = mean; gstat[g][1] = var;
        }
        const float a = invstd * wc, b = bc - mean * a;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            if (off[k] < 0) continue;
            float4 r = v[k];
            r.x = fmaf(r.x, a, b); r.y = fmaf(r.y, a, b); r.z = fmaf(r.z, a, b); r.w = fmaf(r.w, a, b);
            if (residual) { const float4 z = ld4(residual + off[k]); r.x += z.x; r.y += z.y; r.z += z.z; r.w += z.w; }
            if (relu) { r.x = r.x > 0.f ? r.x : 0.f; r.y = r.y > 0.f ? r.y : 0.f; r.z = r.z > 0.f ? r.z : 0.f; r.w = r.w > 0.f ? r.w : 0.f; }
            st4(y + off[k], r);
        }
        __syncthreads();                                  // sh_shift is rewritten by the next group
    }
    if (threadIdx.x == 0 && running_mean) {
        float rm = running_mean[c], rv = running_var[c];
        for (int g = 0; g < G; ++g) {
            const float unbiased = M > 1.f ? gstat[g][1] * (M / (M - 1.f)) : gstat[g][1];
            rm = (1.f - momentum) * rm + momentum * gstat[g][0];
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
}

__global__ void __launch_bounds__(NT) k_bn_bwd_small(const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ gy, const float* __restrict__ weight,
                                                     const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                     float* __restrict__ gx, float* __restrict__ gweight, float* __restrict__ gbias,
                                                     float* __restrict__ g_res, int N, int C, int q, int relu, int accumulate,
                                                     int G, const float* __restrict__ bias) {
    __shared__ float red[8];
    const int c = blockIdx.x, E = N * q;
    const float M = (float)N * (float)(4 * q);
    const float wc = weight ? weight[c] : 1.f;
    const bool remask = relu && y == nullptr;
    float t1 = 0.f, t2 = 0.f;
    for (int g = 0; g < G; ++g) {
        const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
        const float ma = invstd * wc, mb = (bias ? bias[c] : 0.f) - mean * ma;
        float4 d[SMALL_K], xh[SMALL_K];
        long off[SMALL_K];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            const int e = threadIdx.x + k * NT;
            off[k] = -1;
            if (e < E) {
                const int n = e / q, i = e - n * q;
                off[k] = (((long)g * N + n) * C + c) * 4 * q + 4 * i;
                d[k] = ld4(gy + off[k]);
                const float4 xx = ld4(x + off[k]);
                if (remask) {
                    d[k].x = fmaf(xx.x, ma, mb) > 0.f ? d[k].x : 0.f; d[k].y = fmaf(xx.y, ma, mb) > 0.f ? d[k].y : 0.f;
                    d[k].z = fmaf(xx.z, ma, mb) > 0.f ? d[k].z : 0.f; d[k].w = fmaf(xx.w, ma, mb) > 0.f ? d[k].w : 0.f;
                } else if (relu) {
                    const float4 yy = ld4(y + off[k]);
                    d[k].x = yy.x > 0.f ? d[k].x : 0.f; d[k].y = yy.y > 0.f ? d[k].y : 0.f;
                    d[k].z = yy.z > 0.f ? d[k].z : 0.f; d[k].w = yy.w > 0.f ? d[k].w : 0.f;
                }
                xh[k].x = (xx.x - mean) * invstd; xh[k].y = (xx.y - mean) * invstd;
                xh[k].z = (xx.z - mean) * invstd; xh[k].w = (xx.w - mean) * invstd;
                s1 += (d[k].x + d[k].y) + (d[k].z + d[k].w);
                s2 += (d[k].x * xh[k].x + d[k].y * xh[k].y) + (d[k].z * xh[k].z + d[k].w * xh[k].w);
            }
        }
        block_sum2(s1, s2, red);
        t1 += s1; t2 += s2;
        const float kk = wc * invstd, m1 = s1 / M, m2 = s2 / M;
#pragma unroll
        for (int k = 0; k < SMALL_K; ++k) {
            if (off[k] < 0) continue;
            if (g_res) st4(g_res + off[k], d[k]);
            float4 r;
            r.x = kk * (d[k].x - m1 - xh[k].x * m2); r.y = kk * (d[k].y - m1 - xh[k].y * m2);
            r.z = kk * (d[k].z - m1 - xh[k].z * m2); r.w = kk * (d[k].w - m1 - xh[k].w * m2);
            st4(gx + off[k], r);
        }
    }
    if (threadIdx.x == 0) {                              // parameter gradients: sum over the groups, in group order
        if (gbias) gbias[c] = (accumulate ? gbias[c] : 0.f) + t1;
        if (gweight) gweight[c] = (accumulate ? gweight[c] : 0.f) + t2;
    }
}

inline bool bn_small(int Ng, long HW, int groups, bool vec) {
#ifdef FD_ABLATE_NO_BN_SMALL     // timing experiment only: small planes cost nothing (the launches are skipped by the callers)
    (void)Ng; (void)HW; (void)groups; (void)vec;
#endif
    return vec && groups <= 16 && (long)Ng * (HW >> 2) <= (long)NT * SMALL_K;
}

// ---- the stem's tail as ONE pass: relu(bn1(x)) [-> features[0]] -> MaxPool2d(3, 2, 1)  (resnet_encoder.py:95-98) --------------------
// The stem's output is the largest activation of an encoder (64 channels at H/2 x W/2: 189 MB for the pose encoders' 24 stacked
// images at 640x192).  Rounds 1-4 ran apply (read x, write f0), max-pool (read f0, write pooled) and, backward, max-pool adjoint
// (write g_f0), reduction (read g_f0, f0, x) and apply (read g_f0, f0, x, write gx): 11 passes over a tensor of that size.  Here the
// normalised, rectified value is a function of x and two per-channel constants and is recomputed where it is needed:
//   forward   one kernel: pooled + argmax byte from nine taps of max(fma(x, a, b), 0); features[0] is WRITTEN only if somebody reads it
//             (the depth / LiDAR encoders' skip connection - never for the pose encoders);
//   backward  d = relu'(.) * (max-pool adjoint of g_pooled [+ g_f0]) is formed from g_pooled, the argmax bytes and x in both the
//             reduction and the apply pass: 2 reads of x + 1 write of gx.
// One thread owns a 2x2 block of full-resolution pixels = the taps (1..2, 1..2) of pooling window (p, q), as k_maxpool_bwd.
__device__ __forceinline__ float bn_act(float x, float a, float b) { return fmaxf(fmaf(x, a, b), 0.f); }

__global__ void __launch_bounds__(NT) k_bn_relu_pool_fwd(const float* __restrict__ x, const float* __restrict__ weight,
                                                         const float* __restrict__ bias, float* __restrict__ feat,
                                                         float* __restrict__ pooled, uint8_t* __restrict__ idx,
                                                         float* __restrict__ running_mean, float* __restrict__ running_var,
                                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                         const float* __restrict__ part, const float* __restrict__ shifts, int N, int C,
                                                         int H, int W, int splits, float eps, float momentum, int G) {
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    const long HW = (long)H * W;
    const float M = (float)N * (float)HW;
    const BnStat st = bn_finalize(part, shifts, c, C, g, splits, M);
    const float invstd = 1.0f / sqrtf(st.var + eps);
    if (blockIdx.x == 0 && n == g * N && threadIdx.x == 0) { save_mean[g * C + c] = st.mean; save_invstd[g * C + c] = invstd; }
    if (blockIdx.x == 0 && nc < C && threadIdx.x == 0 && running_mean) {      // the G momentum updates in group order (k_bn_apply_train)
        float rm = running_mean[c], rv = running_var[c];
        for (int gg = 0; gg < G; ++gg) {
            const BnStat sg = bn_finalize(part, shifts, c, C, gg, splits, M);
            const float unbiased = M > 1.f ? sg.var * (M / (M - 1.f)) : sg.var;
            rm = (1.f - momentum) * rm + momentum * sg.mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
    const float a = invstd * (weight ? weight[c] : 1.f);
    const float b = (bias ? bias[c] : 0.f) - st.mean * a;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const float* xp = x + (long)nc * HW;
    float* fp = feat ? feat + (long)nc * HW : nullptr;
    for (int r = blockIdx.x * NT + threadIdx.x; r < Ho * Wo; r += gridDim.x * NT) {
        const int ho = r / Wo, wo = r - ho * Wo;
        float best = 0.f;
        int bi = -1;
        float v[3][3];
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
                v[kh][kw] = bn_act(xp[hc * W + wc], a, b);
                const bool take = ok & ((bi < 0) | (v[kh][kw] > best) | (v[kh][kw] != v[kh][kw]));      // first maximum in raster order (k_maxpool_fwd)
                best = take ? v[kh][kw] : best;
                bi = take ? kh * 3 + kw : bi;
            }
        }
        pooled[(long)nc * Ho * Wo + r] = best;
        idx[(long)nc * Ho * Wo + r] = (uint8_t)bi;
        if (fp) {                                                     // this thread's 2x2 block of features[0]
            float* o = fp + (2 * ho) * W + 2 * wo;
            const bool h1 = 2 * ho + 1 < H, w1 = 2 * wo + 1 < W;
            o[0] = v[1][1];
            if (w1) o[1] = v[1][2];
            if (h1) { o[W] = v[2][1]; if (w1) o[W + 1] = v[2][2]; }
        }
    }
}

// Vector form (W % 4 == 0, 16-byte aligned planes - every stem of the step): one thread owns TWO pooling windows (ho, 2j), (ho, 2j+1)
// = a 2 x 4 block of full-resolution pixels: per tap row one float4 + the column left of it instead of six scalars, features[0] as
// float4 rows, pooled / argmax as one 8-byte / 2-byte store; a workgroup covers a quarter of a plane, so the statistics prologue
// (16 partial sums per channel) is paid 4 times per plane instead of 30 times.
__device__ __forceinline__ void pool_take(float v, int tap, bool ok, float& best, int& bi) {
    const bool take = ok & ((bi < 0) | (v > best) | (v != v));
    best = take ? v : best;
    bi = take ? tap : bi;
}
__global__ void __launch_bounds__(NT) k_bn_relu_pool_fwd4(const float* __restrict__ x, const float* __restrict__ weight,
                                                          const float* __restrict__ bias, float* __restrict__ feat,
                                                          float* __restrict__ pooled, uint8_t* __restrict__ idx,
                                                          float* __restrict__ running_mean, float* __restrict__ running_var,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                          const float* __restrict__ part, const float* __restrict__ shifts, int N, int C,
                                                          int H, int W, int splits, float eps, float momentum, int G) {
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    const long HW = (long)H * W;
    const float M = (float)N * (float)HW;
    const BnStat st = bn_finalize(part, shifts, c, C, g, splits, M);
    const float invstd = 1.0f / sqrtf(st.var + eps);
    if (blockIdx.x == 0 && n == g * N && threadIdx.x == 0) { save_mean[g * C + c] = st.mean; save_invstd[g * C + c] = invstd; }
    if (blockIdx.x == 0 && nc < C && threadIdx.x == 0 && running_mean) {
        float rm = running_mean[c], rv = running_var[c];
        for (int gg = 0; gg < G; ++gg) {
            const BnStat sg = bn_finalize(part, shifts, c, C, gg, splits, M);
            const float unbiased = M > 1.f ? sg.var * (M / (M - 1.f)) : sg.var;
            rm = (1.f - momentum) * rm + momentum * sg.mean;
            rv = (1.f - momentum) * rv + momentum * unbiased;
        }
        running_mean[c] = rm; running_var[c] = rv;
    }
    const float a = invstd * (weight ? weight[c] : 1.f);
    const float b = (bias ? bias[c] : 0.f) - st.mean * a;
    const int Ho = (H - 1) / 2 + 1, Wo = W >> 1, Wq = W >> 2;        // W % 4 == 0: Wo even
    const float* xp = x + (long)nc * HW;
    float* fp = feat ? feat + (long)nc * HW : nullptr;
    for (int r = blockIdx.x * NT + threadIdx.x; r < Ho * Wq; r += gridDim.x * NT) {
        const int ho = r / Wq, j = r - ho * Wq;
        float best0 = 0.f, best1 = 0.f;
        int bi0 = -1, bi1 = -1;
        float4 rows[3];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int h = ho * 2 - 1 + kh;
            const bool vh = (unsigned)h < (unsigned)H;
            const int hc = h < 0 ? 0 : (h >= H ? H - 1 : h);
            const float* row = xp + hc * W + 4 * j;
            const float4 q = ld4(row);
            const float l = j > 0 ? row[-1] : 0.f;
            float4 v;
            v.x = bn_act(q.x, a, b); v.y = bn_act(q.y, a, b); v.z = bn_act(q.z, a, b); v.w = bn_act(q.w, a, b);
            rows[kh] = v;
            const float vl = bn_act(l, a, b);
            pool_take(vl, kh * 3 + 0, vh & (j > 0), best0, bi0);      // window 2j: columns 4j-1, 4j, 4j+1
            pool_take(v.x, kh * 3 + 1, vh, best0, bi0);
            pool_take(v.y, kh * 3 + 2, vh, best0, bi0);
            pool_take(v.y, kh * 3 + 0, vh, best1, bi1);               // window 2j+1: columns 4j+1, 4j+2, 4j+3
            pool_take(v.z, kh * 3 + 1, vh, best1, bi1);
            pool_take(v.w, kh * 3 + 2, vh, best1, bi1);
        }
        const long po = (long)nc * Ho * Wo + (long)ho * Wo + 2 * j;
        *reinterpret_cast<float2*>(pooled + po) = make_float2(best0, best1);
        *reinterpret_cast<uint16_t*>(idx + po) = (uint16_t)(bi0 | (bi1 << 8));
        if (fp) {
            st4(fp + (2 * ho) * W + 4 * j, rows[1]);
            if (2 * ho + 1 < H) st4(fp + (2 * ho + 1) * W + 4 * j, rows[2]);
        }
    }
}

// The 2 x 4 block of rows 2p, 2p+1 / columns 4j .. 4j+3: d[8] (row-major) and the x values, from the windows (p .. p+1) x (2j .. 2j+2).
__device__ __forceinline__ void stem_block_grad4(const float* __restrict__ xp, const float* __restrict__ g, const uint8_t* __restrict__ ix,
                                                 const float* __restrict__ gf, int H, int W, int Ho, int Wo, int p, int j, float a, float b,
                                                 float (&o)[8], float (&xs)[8]) {
    const int q = 2 * j;
    const bool vp = p + 1 < Ho, vq = q + 2 < Wo;
    const int p1 = vp ? p + 1 : p, q2 = vq ? q + 2 : q;
    const uint16_t ia = *reinterpret_cast<const uint16_t*>(ix + p * Wo + q), ib = *reinterpret_cast<const uint16_t*>(ix + p1 * Wo + q);
    const int i00 = ia & 255, i01 = ia >> 8, i02 = vq ? ix[p * Wo + q2] : 255;
    const int i10 = vp ? (ib & 255) : 255, i11 = vp ? (ib >> 8) : 255, i12 = (vp & vq) ? ix[p1 * Wo + q2] : 255;
    const float2 ga = *reinterpret_cast<const float2*>(g + p * Wo + q), gb = *reinterpret_cast<const float2*>(g + p1 * Wo + q);
    const float g00 = ga.x, g01 = ga.y, g02 = g[p * Wo + q2], g10 = gb.x, g11 = gb.y, g12 = g[p1 * Wo + q2];
    // block (p, q): k_maxpool_bwd's sums, same order; block (p, q + 1) likewise one window to the right
    o[0] = (i00 == 4 ? g00 : 0.f);
    o[1] = (i00 == 5 ? g00 : 0.f) + (i01 == 3 ? g01 : 0.f);
    o[4] = (i00 == 7 ? g00 : 0.f) + (i10 == 1 ? g10 : 0.f);
    o[5] = (((i00 == 8 ? g00 : 0.f) + (i01 == 6 ? g01 : 0.f)) + (i10 == 2 ? g10 : 0.f)) + (i11 == 0 ? g11 : 0.f);
    o[2] = (i01 == 4 ? g01 : 0.f);
    o[3] = (i01 == 5 ? g01 : 0.f) + (i02 == 3 ? g02 : 0.f);
    o[6] = (i01 == 7 ? g01 : 0.f) + (i11 == 1 ? g11 : 0.f);
    o[7] = (((i01 == 8 ? g01 : 0.f) + (i02 == 6 ? g02 : 0.f)) + (i11 == 2 ? g11 : 0.f)) + (i12 == 0 ? g12 : 0.f);
    const int r0 = 2 * p;
    const bool h1 = r0 + 1 < H;
    const int r1 = h1 ? r0 + 1 : r0;
    const float4 x0 = ld4(xp + r0 * W + 4 * j), x1 = ld4(xp + r1 * W + 4 * j);
    xs[0] = x0.x; xs[1] = x0.y; xs[2] = x0.z; xs[3] = x0.w; xs[4] = x1.x; xs[5] = x1.y; xs[6] = x1.z; xs[7] = x1.w;
    if (gf) {
        const float4 f0 = ld4(gf + r0 * W + 4 * j), f1 = ld4(gf + r1 * W + 4 * j);
        o[0] += f0.x; o[1] += f0.y; o[2] += f0.z; o[3] += f0.w; o[4] += f1.x; o[5] += f1.y; o[6] += f1.z; o[7] += f1.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = fmaf(xs[k], a, b) > 0.f ? o[k] : 0.f;
    if (!h1) { o[4] = o[5] = o[6] = o[7] = 0.f; }
}

// d(2x2 block (p, q)) = relu'(.) * (adjoint of the pooling [+ g_feat]); o[4] = the block's four values (row-major), xs[4] the x values
__device__ __forceinline__ void stem_block_grad(const float* __restrict__ xp, const float* __restrict__ g, const uint8_t* __restrict__ ix,
                                                const float* __restrict__ gf, int H, int W, int Ho, int Wo, int p, int q, float a, float b,
                                                float (&o)[4], float (&xs)[4]) {
    const int p1 = p + 1 < Ho ? p + 1 : p, q1 = q + 1 < Wo ? q + 1 : q;
    const bool vp = p + 1 < Ho, vq = q + 1 < Wo;
    const int i00 = ix[p * Wo + q], i01 = vq ? ix[p * Wo + q1] : 255, i10 = vp ? ix[p1 * Wo + q] : 255, i11 = (vp & vq) ? ix[p1 * Wo + q1] : 255;
    const float g00 = g[p * Wo + q], g01 = g[p * Wo + q1], g10 = g[p1 * Wo + q], g11 = g[p1 * Wo + q1];
    o[0] = (i00 == 4 ? g00 : 0.f);                                    // k_maxpool_bwd's sums, same order
    o[1] = (i00 == 5 ? g00 : 0.f) + (i01 == 3 ? g01 : 0.f);
    o[2] = (i00 == 7 ? g00 : 0.f) + (i10 == 1 ? g10 : 0.f);
    o[3] = (((i00 == 8 ? g00 : 0.f) + (i01 == 6 ? g01 : 0.f)) + (i10 == 2 ? g10 : 0.f)) + (i11 == 0 ? g11 : 0.f);
    const int r0 = 2 * p, c0 = 2 * q;
    const bool h1 = r0 + 1 < H, w1 = c0 + 1 < W;
    const int r1 = h1 ? r0 + 1 : r0, c1 = w1 ? c0 + 1 : c0;
    xs[0] = xp[r0 * W + c0]; xs[1] = xp[r0 * W + c1]; xs[2] = xp[r1 * W + c0]; xs[3] = xp[r1 * W + c1];
    if (gf) { o[0] += gf[r0 * W + c0]; o[1] += gf[r0 * W + c1]; o[2] += gf[r1 * W + c0]; o[3] += gf[r1 * W + c1]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = fmaf(xs[k], a, b) > 0.f ? o[k] : 0.f;
    if (!w1) { o[1] = 0.f; o[3] = 0.f; }                              // cells past the right / lower edge of an odd-sized plane
    if (!h1) { o[2] = 0.f; o[3] = 0.f; }
}

template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_relu_pool_bwd_reduce(const float* __restrict__ x, const float* __restrict__ g_pooled,
                                                                const uint8_t* __restrict__ idx, const float* __restrict__ g_feat,
                                                                const float* __restrict__ weight, const float* __restrict__ bias,
                                                                const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                float* __restrict__ part, int N, int C, int H, int W, int splits) {
    __shared__ float red[4 * 2];
    const int c = blockIdx.x, s = blockIdx.y, g = blockIdx.z;
    const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
    const float a = invstd * (weight ? weight[c] : 1.f), b = (bias ? bias[c] : 0.f) - mean * a;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long HW = (long)H * W;
    const int Wq = VEC ? (W >> 2) : Wb;                              // items per block row: 2 x 4 blocks (VEC) or 2 x 2 blocks
    const int nb = Hb * Wq, per = (nb + splits - 1) / splits;
    const int lo = s * per, hi = lo + per < nb ? lo + per : nb;
    float acc[2] = {0.f, 0.f};
    for (int n = 0; n < N; ++n) {
        const long pl = ((long)g * N + n) * C + c;
        const float* xp = x + pl * HW;
        const float* gp = g_pooled + pl * Ho * Wo;
        const uint8_t* ix = idx + pl * Ho * Wo;
        const float* gf = g_feat ? g_feat + pl * HW : nullptr;
        for (int r = lo + threadIdx.x; r < hi; r += NT) {
            const int p = r / Wq, q = r - p * Wq;
            if (VEC) {
                float o[8], xs[8];
                stem_block_grad4(xp, gp, ix, gf, H, W, Ho, Wo, p, q, a, b, o, xs);
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; k += 2) {
                    t0 += o[k] + o[k + 1];
                    t1 += o[k] * ((xs[k] - mean) * invstd) + o[k + 1] * ((xs[k + 1] - mean) * invstd);
                }
                acc[0] += t0; acc[1] += t1;
            } else {
                float o[4], xs[4];
                stem_block_grad(xp, gp, ix, gf, H, W, Ho, Wo, p, q, a, b, o, xs);
                acc[0] += (o[0] + o[1]) + (o[2] + o[3]);
                acc[1] += (o[0] * ((xs[0] - mean) * invstd) + o[1] * ((xs[1] - mean) * invstd)) +
                          (o[2] * ((xs[2] - mean) * invstd) + o[3] * ((xs[3] - mean) * invstd));
            }
        }
    }
    const float r = fd_block_sum_n<2, 4>(acc, red);
    if (threadIdx.x < 2) part[(((long)g * C + c) * splits + s) * 2 + threadIdx.x] = r;
}

template <bool VEC>
__global__ void __launch_bounds__(NT) k_bn_relu_pool_bwd_apply(const float* __restrict__ x, const float* __restrict__ g_pooled,
                                                               const uint8_t* __restrict__ idx, const float* __restrict__ g_feat,
                                                               const float* __restrict__ weight, const float* __restrict__ bias,
                                                               const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                               float* __restrict__ gx, float* __restrict__ gweight, float* __restrict__ gbias,
                                                               const float* __restrict__ part, int N, int C, int H, int W, int splits,
                                                               int accumulate, int G) {
    const int nc = blockIdx.y, c = nc % C, n = nc / C, g = n / N;
    float s1 = 0.f, s2 = 0.f;
    const long pb = ((long)g * C + c) * splits;
    for (int s = 0; s < splits; ++s) { s1 += part[(pb + s) * 2]; s2 += part[(pb + s) * 2 + 1]; }
    if (blockIdx.x == 0 && nc < C && threadIdx.x == 0) {   // parameter gradients: sum over the groups, in group order (k_bn_bwd_apply)
        float t1 = 0.f, t2 = 0.f;
        for (int gg = 0; gg < G; ++gg) {
            float u1 = 0.f, u2 = 0.f;
            const long qb = ((long)gg * C + c) * splits;
            for (int s = 0; s < splits; ++s) { u1 += part[(qb + s) * 2]; u2 += part[(qb + s) * 2 + 1]; }
            t1 += u1; t2 += u2;
        }
        if (gbias) gbias[c] = (accumulate ? gbias[c] : 0.f) + t1;
        if (gweight) gweight[c] = (accumulate ? gweight[c] : 0.f) + t2;
    }
    const long HW = (long)H * W;
    const float M = (float)N * (float)HW;
    const float mean = save_mean[g * C + c], invstd = save_invstd[g * C + c];
    const float wc = weight ? weight[c] : 1.f;
    const float a = invstd * wc, b = (bias ? bias[c] : 0.f) - mean * a;
    const float k = wc * invstd, m1 = s1 / M, m2 = s2 / M;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const float* xp = x + (long)nc * HW;
    const float* gp = g_pooled + (long)nc * Ho * Wo;
    const uint8_t* ix = idx + (long)nc * Ho * Wo;
    const float* gf = g_feat ? g_feat + (long)nc * HW : nullptr;
    float* op = gx + (long)nc * HW;
    if (VEC) {
        const int Wq = W >> 2;
        for (int r = blockIdx.x * NT + threadIdx.x; r < Hb * Wq; r += gridDim.x * NT) {
            const int p = r / Wq, j = r - p * Wq;
            float o[8], xs[8];
            stem_block_grad4(xp, gp, ix, gf, H, W, Ho, Wo, p, j, a, b, o, xs);
            float v[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) v[t] = k * (o[t] - m1 - ((xs[t] - mean) * invstd) * m2);
            st4(op + (2 * p) * W + 4 * j, make_float4(v[0], v[1], v[2], v[3]));
            if (2 * p + 1 < H) st4(op + (2 * p + 1) * W + 4 * j, make_float4(v[4], v[5], v[6], v[7]));
        }
        return;
    }
    for (int r = blockIdx.x * NT + threadIdx.x; r < Hb * Wb; r += gridDim.x * NT) {
        const int p = r / Wb, q = r - p * Wb;
        float o[4], xs[4];
        stem_block_grad(xp, gp, ix, gf, H, W, Ho, Wo, p, q, a, b, o, xs);
        float* w_ = op + (2 * p) * W + 2 * q;
        const bool h1 = 2 * p + 1 < H, w1 = 2 * q + 1 < W;
        w_[0] = k * (o[0] - m1 - ((xs[0] - mean) * invstd) * m2);
        if (w1) w_[1] = k * (o[1] - m1 - ((xs[1] - mean) * invstd) * m2);
        if (h1) {
            w_[W] = k * (o[2] - m1 - ((xs[2] - mean) * invstd) * m2);
            if (w1) w_[W + 1] = k * (o[3] - m1 - ((xs[3] - mean) * invstd) * m2);
        }
    }
}

// vector path of the stem-tail kernels: rows of 4-float groups, planes 16-byte aligned (pooled rows then hold an even number of floats)
inline bool stem_vec_ok(int W, const void* a, const void* b, const void* c, const void* d, const void* e) {
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; };
    return (W & 3) == 0 && al(a) && al(b) && al(c) && al(d) && al(e);
}
inline int plane_blocks(long HW) {
    long b = (HW + NT * 4 - 1) / (NT * 4);
    return (int)(b < 1 ? 1 : (b > 64 ? 64 : b));
}

}  // namespace

bool bn_small_slabs_ok(int N, int C, int H, int W, int groups) {
    if (groups < 1 || groups > 16 || N % groups || (W & 3) || (H & 1) || C < 1) return false;
    return bn_small(N / groups, (long)H * W, groups, true);
}
int bn_small_slabs_launch(const float* slabs, long slab_stride, int ksplit, float* y, const BnAfterConv& bn, int N, int C, int H, int W,
                          hipStream_t st) {
    FD_REQUIRE(slabs && y && bn.out && bn.save_mean && bn.save_invstd && ksplit >= 1, "conv + BatchNorm: NULL argument");
    FD_REQUIRE(bn_small_slabs_ok(N, C, H, W, bn.groups), "conv + BatchNorm: shape is not a small-plane BatchNorm behind a slab convolution");
    FD_REQUIRE((((uintptr_t)slabs | (uintptr_t)y | (uintptr_t)bn.out | (uintptr_t)bn.residual) & 15) == 0 && (slab_stride & 3) == 0,
               "conv + BatchNorm: 16-byte aligned tensors needed");
    hipLaunchKernelGGL(k_bn_train_small_slabs, dim3(C), dim3(NT), 0, st, slabs, slab_stride, ksplit, y, bn.weight, bn.bias, bn.residual, bn.out,
                       bn.running_mean, bn.running_var, bn.save_mean, bn.save_invstd, N / bn.groups, C, H, W, bn.eps, bn.momentum, bn.relu, bn.groups);
    FD_LAUNCH_CHECK("conv + BatchNorm (slabs)");
    return 0;
}

extern "C" long fd_bn_ws_floats(int N, int C, int H, int W, int groups) {
    if (groups < 1 || N % groups) return 0;
    return (long)groups * C * (bn_splits(N / groups, C, (long)H * W, groups) * 2 + 1);      // partial sums + one shift per (group, channel)
}

extern "C" int fd_bn_train_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                               float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* ws,
                               int N, int C, int H, int W, int groups, float eps, float momentum, int relu, void* stream) {
    FD_REQUIRE(x && y && save_mean && save_invstd && ws && N > 0 && C > 0 && H > 0 && W > 0, "fd_bn_train_fwd: bad args");
    FD_REQUIRE(groups >= 1 && N % groups == 0, "fd_bn_train_fwd: batch %d is not divisible into %d groups", N, groups);
    FD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "fd_bn_train_fwd: running stats must come in pairs");
#ifdef FD_ABLATE_NO_BN          // timing experiment only (wrong results): what the step would gain if BatchNorm cost nothing
    return 0;
#endif
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int Ng = N / groups;
    const int sp = bn_splits(Ng, C, HW, groups);
    const bool vec = bn_vec_ok(HW, x, y, residual, nullptr, nullptr);
    if (bn_small(Ng, HW, groups, vec)) {
#ifdef FD_ABLATE_NO_BN_SMALL
        return 0;
#endif
        hipLaunchKernelGGL(k_bn_train_small, dim3(C), dim3(NT), 0, st, x, weight, bias, residual, y, running_mean, running_var,
                           save_mean, save_invstd, Ng, C, (int)(HW >> 2), eps, momentum, relu, groups);
        FD_LAUNCH_CHECK("fd_bn_train_fwd(small)");
        return 0;
    }
    auto stats = vec ? k_bn_stats<true> : k_bn_stats<false>;
    auto apply = vec ? k_bn_apply_train<true> : k_bn_apply_train<false>;
#ifndef FD_ABLATE_NO_BN_STATS    // timing experiment only: the apply pass then normalises with whatever the workspace holds
    float* shifts = ws + (long)groups * C * sp * 2;              // [group][channel], behind the partial sums
    hipLaunchKernelGGL(stats, dim3(C, sp, groups), dim3(NT), 0, st, x, ws, shifts, Ng, C, HW, sp);
    FD_LAUNCH_CHECK("fd_bn_train_fwd(stats)");
#endif
    hipLaunchKernelGGL(apply, dim3(plane_blocks(HW), N * C), dim3(NT), 0, st, x, weight, bias, residual, y,
                       running_mean, running_var, save_mean, save_invstd, ws, ws + (long)groups * C * sp * 2, Ng, C, HW, sp, eps, momentum,
                       relu, groups);
    FD_LAUNCH_CHECK("fd_bn_train_fwd(apply)");
    return 0;
}

extern "C" int fd_bn_train_fwd_parts(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                                     float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                     const float* conv_part, int slots, int N, int C, int H, int W, int groups, float eps,
                                     float momentum, int relu, void* stream) {
    FD_REQUIRE(x && y && save_mean && save_invstd && conv_part && slots > 0 && N > 0 && C > 0 && H > 0 && W > 0,
               "fd_bn_train_fwd_parts: bad args");
    FD_REQUIRE(groups >= 1 && groups <= 16 && N % groups == 0, "fd_bn_train_fwd_parts: batch %d is not divisible into %d groups (<= 16)", N, groups);
    FD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "fd_bn_train_fwd_parts: running stats must come in pairs");
    const long HW = (long)H * W;
    const bool vec = bn_vec_ok(HW, x, y, residual, nullptr, nullptr);
    auto apply = vec ? k_bn_apply_parts<true> : k_bn_apply_parts<false>;
    // every workgroup first reduces the N / groups * slots partial sums of its (group, channel): give it a whole plane (up to
    // 16 float4 per thread) so that this prologue stays a small fraction of the bytes it moves
    const int bx = (int)((HW + 16 * 4 * NT - 1) / (16 * 4 * NT));
    hipLaunchKernelGGL(apply, dim3(bx < 1 ? 1 : (bx > 64 ? 64 : bx), N * C), dim3(NT), 0, (hipStream_t)stream, x, weight, bias, residual, y,
                       running_mean, running_var, save_mean, save_invstd, conv_part, N / groups, C, HW, slots, eps, momentum, relu,
                       groups);
    FD_LAUNCH_CHECK("fd_bn_train_fwd_parts");
    return 0;
}

extern "C" int fd_bn_eval_fwd(const float* x, const float* weight, const float* bias, const float* residual, float* y,
                              const float* running_mean, const float* running_var, int N, int C, int H, int W, float eps,
                              int relu, void* stream) {
    FD_REQUIRE(x && y && running_mean && running_var && N > 0 && C > 0 && H > 0 && W > 0, "fd_bn_eval_fwd: bad args");
    const long HW = (long)H * W;
    hipLaunchKernelGGL(k_bn_apply_eval, dim3(plane_blocks(HW), N * C), dim3(NT), 0, (hipStream_t)stream, x, weight, bias,
                       residual, y, running_mean, running_var, C, HW, eps, relu);
    FD_LAUNCH_CHECK("fd_bn_eval_fwd");
    return 0;
}

static int bn_train_bwd_impl(const float* x, const float* y, const float* gy, const float* weight, const float* bias,
                             const float* save_mean, const float* save_invstd, float* gx, float* gweight, float* gbias,
                             float* g_residual, float* ws, int N, int C, int H, int W, int groups, int relu, int accumulate,
                             void* stream) {
    FD_REQUIRE(x && gy && save_mean && save_invstd && gx && ws && N > 0 && C > 0 && H > 0 && W > 0,
               "fd_bn_train_bwd: bad args");
    FD_REQUIRE(groups >= 1 && N % groups == 0, "fd_bn_train_bwd: batch %d is not divisible into %d groups", N, groups);
#ifdef FD_ABLATE_NO_BN
    return 0;
#endif
    FD_REQUIRE(!relu || y || !g_residual, "fd_bn_train_bwd: a BatchNorm with a residual input needs the forward output for its ReLU mask");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int Ng = N / groups;
    const int sp = bn_splits(Ng, C, HW, groups);
    const bool vec = bn_vec_ok(HW, x, y, gy, gx, g_residual);
    if (bn_small(Ng, HW, groups, vec)) {
#ifdef FD_ABLATE_NO_BN_SMALL
        return 0;
#endif
        hipLaunchKernelGGL(k_bn_bwd_small, dim3(C), dim3(NT), 0, st, x, y, gy, weight, save_mean, save_invstd, gx, gweight, gbias,
                           g_residual, Ng, C, (int)(HW >> 2), relu, accumulate, groups, bias);
        FD_LAUNCH_CHECK("fd_bn_train_bwd(small)");
        return 0;
    }
    auto reduce = vec ? k_bn_bwd_reduce<true> : k_bn_bwd_reduce<false>;
    auto apply = vec ? k_bn_bwd_apply<true> : k_bn_bwd_apply<false>;
#ifndef FD_ABLATE_NO_BN_STATS
    hipLaunchKernelGGL(reduce, dim3(C, sp, groups), dim3(NT), 0, st, x, y, gy, save_mean, save_invstd, ws, Ng, C, HW,
                       sp, relu, weight, bias);
    FD_LAUNCH_CHECK("fd_bn_train_bwd(reduce)");
#endif
    hipLaunchKernelGGL(apply, dim3(plane_blocks(HW), N * C), dim3(NT), 0, st, x, y, gy, weight, save_mean,
                       save_invstd, gx, gweight, gbias, g_residual, ws, Ng, C, HW, sp, relu, accumulate, groups, bias);
    FD_LAUNCH_CHECK("fd_bn_train_bwd(apply)");
    return 0;
}

extern "C" int fd_bn_train_bwd(const float* x, const float* y, const float* gy, const float* weight, const float* save_mean,
                               const float* save_invstd, float* gx, float* gweight, float* gbias, float* g_residual, float* ws, int N,
                               int C, int H, int W, int groups, int relu, int accumulate, void* stream) {
    FD_REQUIRE(!relu || y, "fd_bn_train_bwd: the forward output is needed for the ReLU mask (or use fd_bn_train_bwd_remask)");
    return bn_train_bwd_impl(x, y, gy, weight, nullptr, save_mean, save_invstd, gx, gweight, gbias, g_residual, ws, N, C, H, W, groups, relu,
                             accumulate, stream);
}

extern "C" int fd_bn_train_bwd_remask(const float* x, const float* gy, const float* weight, const float* bias, const float* save_mean,
                                      const float* save_invstd, float* gx, float* gweight, float* gbias, float* ws, int N, int C, int H,
                                      int W, int groups, int accumulate, void* stream) {
    return bn_train_bwd_impl(x, nullptr, gy, weight, bias, save_mean, save_invstd, gx, gweight, gbias, nullptr, ws, N, C, H, W, groups, 1,
                             accumulate, stream);
}

extern "C" int fd_bn_relu_maxpool_fwd(const float* x, const float* weight, const float* bias, float* feat, float* pooled, uint8_t* idx,
                                      float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* ws, int N, int C,
                                      int H, int W, int groups, float eps, float momentum, void* stream) {
    FD_REQUIRE(x && pooled && idx && save_mean && save_invstd && ws && N > 0 && C > 0 && H > 0 && W > 0, "fd_bn_relu_maxpool_fwd: bad args");
    FD_REQUIRE(groups >= 1 && N % groups == 0, "fd_bn_relu_maxpool_fwd: batch %d is not divisible into %d groups", N, groups);
    FD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "fd_bn_relu_maxpool_fwd: running stats must come in pairs");
    FD_REQUIRE((long)H * W < (1L << 30), "fd_bn_relu_maxpool_fwd: plane too large");
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int Ng = N / groups;
    const int sp = bn_splits(Ng, C, HW, groups);
    const bool vec = bn_vec_ok(HW, x, nullptr, nullptr, nullptr, nullptr);
    float* shifts = ws + (long)groups * C * sp * 2;
    auto stats = vec ? k_bn_stats<true> : k_bn_stats<false>;
    hipLaunchKernelGGL(stats, dim3(C, sp, groups), dim3(NT), 0, st, x, ws, shifts, Ng, C, HW, sp);
    FD_LAUNCH_CHECK("fd_bn_relu_maxpool_fwd(stats)");
    const long po = (long)((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    if (stem_vec_ok(W, x, feat, pooled, nullptr, idx)) {
        const long items = (long)((H - 1) / 2 + 1) * (W / 4);
        const long bx4 = (items + 4 * NT - 1) / (4 * NT);            // ~4 items per thread: the statistics prologue once per quarter plane
        hipLaunchKernelGGL(k_bn_relu_pool_fwd4, dim3((unsigned)(bx4 < 1 ? 1 : (bx4 > 64 ? 64 : bx4)), N * C), dim3(NT), 0, st, x, weight, bias,
                           feat, pooled, idx, running_mean, running_var, save_mean, save_invstd, ws, shifts, Ng, C, H, W, sp, eps, momentum, groups);
        FD_LAUNCH_CHECK("fd_bn_relu_maxpool_fwd(vec)");
        return 0;
    }
    long bx = (po + NT - 1) / NT;
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL(k_bn_relu_pool_fwd, dim3((unsigned)bx, N * C), dim3(NT), 0, st, x, weight, bias, feat, pooled, idx, running_mean,
                       running_var, save_mean, save_invstd, ws, shifts, Ng, C, H, W, sp, eps, momentum, groups);
    FD_LAUNCH_CHECK("fd_bn_relu_maxpool_fwd");
    return 0;
}

extern "C" int fd_bn_relu_maxpool_bwd(const float* x, const float* g_pooled, const uint8_t* idx, const float* g_feat, const float* weight,
                                      const float* bias, const float* save_mean, const float* save_invstd, float* gx, float* gweight,
                                      float* gbias, float* ws, int N, int C, int H, int W, int groups, int accumulate, void* stream) {
    FD_REQUIRE(x && g_pooled && idx && save_mean && save_invstd && gx && ws && N > 0 && C > 0 && H > 0 && W > 0, "fd_bn_relu_maxpool_bwd: bad args");
    FD_REQUIRE(groups >= 1 && N % groups == 0, "fd_bn_relu_maxpool_bwd: batch %d is not divisible into %d groups", N, groups);
    hipStream_t st = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int Ng = N / groups;
    const int sp = bn_splits(Ng, C, HW, groups);
    const bool vec = stem_vec_ok(W, x, g_feat, g_pooled, gx, idx);
    hipLaunchKernelGGL((vec ? k_bn_relu_pool_bwd_reduce<true> : k_bn_relu_pool_bwd_reduce<false>), dim3(C, sp, groups), dim3(NT), 0, st, x,
                       g_pooled, idx, g_feat, weight, bias, save_mean, save_invstd, ws, Ng, C, H, W, sp);
    FD_LAUNCH_CHECK("fd_bn_relu_maxpool_bwd(reduce)");
    const long nb = vec ? (long)((H + 1) / 2) * (W / 4) : (long)((H + 1) / 2) * ((W + 1) / 2);
    long bx = vec ? (nb + 4 * NT - 1) / (4 * NT) : (nb + NT - 1) / NT;
    bx = bx < 1 ? 1 : (bx > 64 ? 64 : bx);
    hipLaunchKernelGGL((vec ? k_bn_relu_pool_bwd_apply<true> : k_bn_relu_pool_bwd_apply<false>), dim3((unsigned)bx, N * C), dim3(NT), 0, st, x,
                       g_pooled, idx, g_feat, weight, bias, save_mean, save_invstd, gx, gweight, gbias, ws, Ng, C, H, W, sp, accumulate, groups);
    FD_LAUNCH_CHECK("fd_bn_relu_maxpool_bwd(apply)");
    return 0;
}
