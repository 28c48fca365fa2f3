// FP32 implicit-GEMM convolution on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32, k-ordered
// fmaf chain) — forward, data-gradient and weight-gradient for every conv of the hot path:
//   ResNet trunk   networks/resnet_encoder.py:92-103  (7x7 s2, 3x3 s1/s2, 1x1 s1/s2; zero pad)
//   DepthDecoder   networks/depth_decoder.py:63-96     (3x3 reflect pad + bias + ELU / sigmoid)
//   PoseDecoder    networks/pose_decoder.py:29-51      (1x1 / 3x3 + bias + ReLU)
//
// One "gather GEMM" kernel does forward AND data-gradient:  D[m][p] = sum_k A[m][k] * G[k][p]
//   A   dense row-major [M][K] matrix in HBM (weights, or a re-laid-out copy made by the prep kernels)
//   G   never materialised: k = (c, a, b) indexes channel c and tap (a,b); p = (n, y, x) indexes a pixel of
//       the GEMM-N domain; G[k][p] = X[n][c][y*sy+oy+a*da][x*sx+ox+b*db]  (zero or reflect outside)
//   D   written through an affine pixel map (so stride-2 dgrad parity classes scatter into dX directly).
// NCHW keeps pixels contiguous, so both the G loads and the D stores are coalesced along the 64 lanes.
// Tiles: workgroup = WAVES_M x WAVES_N waves, each wave owns WM x WN accumulators of 32x32, K-chunk 16,
// register-staged double-buffered LDS (one barrier per chunk).
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include "conv_limb.h"
#include <stdio.h>
#include "conv_narrow.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;

struct GemmArgs {
    const float* A; const float* X; float* Y; const float* bias;
    int M, K;
    int Nb, C, Hi, Wi;
    int NY, NX;
    int sy, oy, da, sx, ox, db;
    int pad_mode;   // 0 zero, 1 reflect
    long out_ns, out_cs;
    int out_w, osy, ooy, osx, oox;
    int act;        // 0 none, 1 relu, 2 elu, 3 sigmoid, 4 tanh
    int in_norm;    // conv1: (x - 0.45) / 0.225 on in-bounds taps (resnet_encoder.py:94)
    int xcd_swizzle;
};

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    if (act == 4) return tanhf(v);
    return v;
}

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

template <int TA, int TB, int WAVES_M, int WAVES_N, int WM, int WN, bool NORM>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_gather_gemm(GemmArgs g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    constexpr int LDA = BM + 2, LDB = BN;
    constexpr int RP = NT / BN;            // k rows covered per pass of the G loader
    constexpr int NB_LOAD = BK / RP;       // G elements per thread per chunk
    constexpr int MP = NT / BK;            // m rows covered per pass of the A loader
    constexpr int NA_LOAD = BM / MP;
    static_assert(NT % BN == 0 && BK % RP == 0 && BM % MP == 0, "tile/loader mismatch");
    __shared__ float sA[2][BK * LDA];
    __shared__ float sB[2][BK * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;

    // block -> tile (optionally XCD-aware: consecutive pixel tiles stay on one XCD / one L2)
    int bx = blockIdx.x;
    if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const int m0 = blockIdx.y * BM;
    const long p0 = (long)bx * BN;
    const long plane = (long)g.NY * g.NX, Np = (long)g.Nb * plane;
    const long chw = (long)g.Hi * g.Wi;

    // ---- G loader: this thread always fetches pixel column jn, k rows kr + RP*i
    const int jn = tid % BN;
    const int kr = __builtin_amdgcn_readfirstlane(tid / BN);
    const long pg = p0 + jn;
    const bool pvalid = pg < Np;
    int ry0 = 0, cx0 = 0;
    unsigned nbase = 0u;
    {
        const long pp = pvalid ? pg : 0;
        const int n = (int)(pp / plane);
        const int rem = (int)(pp - (long)n * plane);
        const int y = rem / g.NX, x = rem - y * g.NX;
        ry0 = y * g.sy + g.oy; cx0 = x * g.sx + g.ox;
        nbase = (unsigned)n * (unsigned)g.C * (unsigned)chw;
    }
    // ---- A loader: k column ka, m rows ma + MP*i
    const int ka = tid % BK, ma = tid / BK;
    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.A), rsX = fd_make_rsrc(g.X);
    const bool refl = g.pad_mode == 1;

    // Raw buffer loads (32-bit byte offsets, out-of-range = 0.0f): padding taps, k >= K, rows >= M and pixels past the end
    // need no selects; only the NORM variant keeps a validity mask because its padding must stay 0 AFTER the affine map.
    float ra[NA_LOAD], rb[NB_LOAD];
    unsigned okmask = 0u;
    int k0 = 0;                                        // first k of the chunk being fetched
    auto load_a = [&](int i) __attribute__((always_inline)) {
        const int m = m0 + ma + MP * i, k = k0 + ka;
        ra[i] = fd_ldg32(rsA, (m < g.M) & (k < g.K) ? 4u * ((unsigned)m * (unsigned)g.K + (unsigned)k) : FD_OOB);
    };
    auto load_b = [&](int i) __attribute__((always_inline)) {
        const int k = k0 + kr + RP * i;                // wave-uniform
        const int c = k / (TA * TB), t = k - c * (TA * TB);
        const int ta = t / TB, tb = t - ta * TB;
        int r = ry0 + ta * g.da, cc = cx0 + tb * g.db;
        const bool inside = ((unsigned)r < (unsigned)g.Hi) & ((unsigned)cc < (unsigned)g.Wi);
        const int rr = reflect_idx(r, g.Hi), cr = reflect_idx(cc, g.Wi);
        r = refl ? rr : r; cc = refl ? cr : cc;
        const bool ok = pvalid & (k < g.K) & (refl | inside);
        rb[i] = fd_ldg32(rsX, ok ? 4u * (nbase + (unsigned)c * (unsigned)chw + (unsigned)(r * g.Wi + cc)) : FD_OOB);
        if (NORM) okmask = (okmask & ~(1u << i)) | (ok ? (1u << i) : 0u);
    };
    auto store_a = [&](int buf, int i) __attribute__((always_inline)) { sA[buf][ka * LDA + ma + MP * i] = ra[i]; };
    auto store_b = [&](int buf, int i) __attribute__((always_inline)) {
        float v = rb[i];
        if (NORM) v = ((okmask >> i) & 1u) ? (v - 0.45f) / 0.225f : 0.f;          // resnet_encoder.py:94, padding stays 0
        sB[buf][(kr + RP * i) * LDB + jn] = v;
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunk = (g.K + BK - 1) / BK;
    constexpr int NK = BK / 2, LS = NK / 2;           // first LS k-steps issue the next chunk's loads, the last LS store them
#pragma unroll
    for (int i = 0; i < NA_LOAD; ++i) load_a(i);
#pragma unroll
    for (int i = 0; i < NB_LOAD; ++i) load_b(i);
#pragma unroll
    for (int i = 0; i < NA_LOAD; ++i) store_a(0, i);
#pragma unroll
    for (int i = 0; i < NB_LOAD; ++i) store_b(0, i);
    __syncthreads();
    const int arow = lane >> 5, acol = lane & 31;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int cur = ch & 1;
        k0 = (ch + 1) * BK;                            // past the end: k >= K, every load out of range
        const float* pa = &sA[cur][arow * LDA + wave_m * 32 * WM + acol];
        const float* pb = &sB[cur][arow * LDB + wave_n * 32 * WN + acol];
        float av[2][WM], bv[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) av[0][i] = pa[i * 32];
#pragma unroll
        for (int j = 0; j < WN; ++j) bv[0][j] = pb[j * 32];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int cb = kk & 1, nb = cb ^ 1;
            if (kk + 1 < NK) {
#pragma unroll
                for (int i = 0; i < WM; ++i) av[nb][i] = pa[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
                for (int j = 0; j < WN; ++j) bv[nb][j] = pb[(kk + 1) * 2 * LDB + j * 32];
            }
            if (kk < LS) {
#pragma unroll
                for (int i = 0; i < NA_LOAD; ++i) if ((i * LS) / NA_LOAD == kk) load_a(i);
#pragma unroll
                for (int i = 0; i < NB_LOAD; ++i) if ((i * LS) / NB_LOAD == kk) load_b(i);
            } else if (kk >= NK - LS) {
#pragma unroll
                for (int i = 0; i < NA_LOAD; ++i) if ((i * LS) / NA_LOAD == kk - (NK - LS)) store_a(cur ^ 1, i);
#pragma unroll
                for (int i = 0; i < NB_LOAD; ++i) if ((i * LS) / NB_LOAD == kk - (NK - LS)) store_b(cur ^ 1, i);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][i], bv[cb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: bias + activation, affine pixel map.  C/D layout of 32x32 MFMA:
    //      col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const long p = p0 + wave_n * 32 * WN + j * 32 + acol;
        if (p >= Np) continue;
        const int n = (int)(p / plane);
        const int rem = (int)(p - (long)n * plane);
        const int y = rem / g.NX, x = rem - y * g.NX;
        float* yo = g.Y + (long)n * g.out_ns + (long)(y * g.osy + g.ooy) * g.out_w + (x * g.osx + g.oox);
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                if (m < g.M) {
                    float v = acc[i][j][r];
                    if (g.bias) v += g.bias[m];
                    yo[(long)m * g.out_cs] = apply_act(v, g.act);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient:  dW[m][j] = sum_p dY[m][p] * G[j][p]   (j = (c,a,b) as above, p over the OUTPUT pixels
// of the forward conv), split over the pixel axis; partial slabs are reduced in a fixed order.
struct WgradArgs {
    const float* dY; const float* X; float* out;   // out: dW (splits == 1) or slabs [splits][M][J]
    int M, J;
    int Nb, C, Hi, Wi;
    int NY, NX;
    int sy, oy, da, sx, ox, db;
    int pad_mode, in_norm;
    long dy_ns, dy_cs;      // dY[n*dy_ns + m*dy_cs + y*NX + x]
    long pix_per_split;
};

// Loader design (same recipe as conv_fast.hip): raw buffer loads with 32-bit byte offsets, out-of-range = 0.0f; every
// per-row quantity (dY row offset, the (channel, tap) decode of a G row) is fixed per thread and precomputed; the pixel
// decode advances incrementally; loads of chunk ch+1 / their LDS stores are interleaved with the MFMAs of chunk ch.
template <int TA, int TB, int WAVES_M, int WAVES_N, int WM, int WN, bool REFL, bool NORM>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_wgrad(WgradArgs g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    constexpr int BP = 32;                       // pixels (GEMM-K) per chunk
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int RPW = NT / BP;                 // rows (m or j) covered per pass
    constexpr int NA_LOAD = BM / RPW, NB_LOAD = BN / RPW;
    static_assert(BM % RPW == 0 && BN % RPW == 0, "tile/loader mismatch");
    __shared__ float sA[2][BP * LDA];
    __shared__ float sB[2][BP * LDB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int m0 = blockIdx.y * BM, j0 = blockIdx.x * BN;
    const int plane = g.NY * g.NX;
    const long Np = (long)g.Nb * plane;
    const unsigned chw = (unsigned)(g.Hi * g.Wi);
    const long pbeg = (long)blockIdx.z * g.pix_per_split;
    long pend = pbeg + g.pix_per_split;
    if (pend > Np) pend = Np;

    const int pl = tid % BP, rw = tid / BP;      // pixel within chunk, first row handled
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.dY), rsX = fd_make_rsrc(g.X);
    const int nrow = g.M - m0 < BM ? g.M - m0 : BM;
    // dY rows past M are clamped to the last valid row: their products land in accumulator rows that are never stored
    unsigned rowa[NA_LOAD];
#pragma unroll
    for (int i = 0; i < NA_LOAD; ++i) {
        const int r = rw + RPW * i < nrow ? rw + RPW * i : nrow - 1;
        rowa[i] = 4u * (unsigned)(m0 + r) * (unsigned)g.dy_cs;
    }
    // G rows j = (channel, tap): byte offset of the (channel, tap) relative to the pixel's gather origin, and the tap
    // displacement for the bounds / reflect logic.  Rows >= J are clamped to row J-1 (their output columns are never stored).
    unsigned joff[NB_LOAD];
    int jro[NB_LOAD], jco[NB_LOAD];
#pragma unroll
    for (int i = 0; i < NB_LOAD; ++i) {
        const int j = j0 + rw + RPW * i < g.J ? j0 + rw + RPW * i : g.J - 1;
        const int c = j / (TA * TB), t = j - c * (TA * TB);
        const int ta = t / TB, tb = t - ta * TB;
        jro[i] = ta * g.da; jco[i] = tb * g.db;
        if (REFL) joff[i] = 4u * (unsigned)c * chw;
        else joff[i] = 4u * ((unsigned)c * chw + (unsigned)(jro[i] * g.Wi + jco[i]));
    }

    float ra[NA_LOAD], rb[NB_LOAD];
    unsigned offa = FD_OOB, xbase = FD_OOB, okmask = 0u;
    int ry0 = 0, cx0 = 0;
    int pn, prem;
    { const long p = pbeg + pl; pn = (int)(p / plane); prem = (int)(p - (long)pn * plane); }
    long pcur = pbeg + pl;
    const float inv_nx = 1.0f / (float)g.NX;
    auto prep_chunk = [&]() __attribute__((always_inline)) {      // pixels >= pend: everything out of range (zeros)
        const bool pv = pcur < pend;
        int y = (int)(((float)prem + 0.5f) * inv_nx);             // estimate within +-1 for planes < 2^23; fixed up below
        int x = prem - y * g.NX;
        if (x < 0) { --y; x += g.NX; }
        if (x >= g.NX) { ++y; x -= g.NX; }
        offa = pv ? 4u * ((unsigned)pn * (unsigned)g.dy_ns + (unsigned)prem) : FD_OOB;
        ry0 = y * g.sy + g.oy; cx0 = x * g.sx + g.ox;
        // zero padding: origin of the gather window (may lie outside the image: the sum with joff is used only in bounds)
        if (REFL) xbase = pv ? 4u * (unsigned)pn * (unsigned)g.C * chw : FD_OOB;
        else xbase = 4u * ((unsigned)pn * (unsigned)g.C * chw + (unsigned)(ry0 * g.Wi + cx0));
        if (!REFL && !pv) ry0 = -(1 << 20);                       // fails every bounds test below
        pcur += BP; prem += BP;
        while (prem >= plane) { prem -= plane; ++pn; }
        okmask = 0u;
    };
    auto load_a = [&](int i) __attribute__((always_inline)) { ra[i] = fd_ldg32(rsY, offa + rowa[i]); };
    auto load_b = [&](int i) __attribute__((always_inline)) {
        const int r = ry0 + jro[i], cc = cx0 + jco[i];
        unsigned off;
        bool ok;
        if (REFL) {
            const int rr = reflect_idx(r, g.Hi), cr = reflect_idx(cc, g.Wi);
            off = xbase + joff[i] + 4u * (unsigned)(rr * g.Wi + cr);          // xbase carries FD_OOB for pixels past the end
            ok = true;
        } else {
            ok = ((unsigned)r < (unsigned)g.Hi) & ((unsigned)cc < (unsigned)g.Wi);
            off = ok ? xbase + joff[i] : FD_OOB;
        }
        if (NORM) okmask |= ok ? (1u << i) : 0u;
        rb[i] = fd_ldg32(rsX, off);
    };
    auto store_a = [&](int buf, int i) __attribute__((always_inline)) { sA[buf][pl * LDA + rw + RPW * i] = ra[i]; };
    auto store_b = [&](int buf, int i) __attribute__((always_inline)) {
        float v = rb[i];
        if (NORM) v = ((okmask >> i) & 1u) ? (v - 0.45f) / 0.225f : 0.f;      // resnet_encoder.py:94, padding stays 0
        sB[buf][pl * LDB + rw + RPW * i] = v;
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunk = pend > pbeg ? (int)((pend - pbeg + BP - 1) / BP) : 0;
    constexpr int NK = BP / 2, HS = NK / 2;
    const int arow = lane >> 5, acol = lane & 31;
    if (nchunk > 0) {
        prep_chunk();
#pragma unroll
        for (int i = 0; i < NA_LOAD; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < NB_LOAD; ++i) load_b(i);
#pragma unroll
        for (int i = 0; i < NA_LOAD; ++i) store_a(0, i);
#pragma unroll
        for (int i = 0; i < NB_LOAD; ++i) store_b(0, i);
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            prep_chunk();
            const float* pa = &sA[cur][arow * LDA + wave_m * 32 * WM + acol];
            const float* pb = &sB[cur][arow * LDB + wave_n * 32 * WN + acol];
            float av[2][WM], bv[2][WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) av[0][i] = pa[i * 32];
#pragma unroll
            for (int j = 0; j < WN; ++j) bv[0][j] = pb[j * 32];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk + 1 < NK) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) av[nb][i] = pa[(kk + 1) * 2 * LDA + i * 32];
#pragma unroll
                    for (int j = 0; j < WN; ++j) bv[nb][j] = pb[(kk + 1) * 2 * LDB + j * 32];
                }
                if (kk < HS) {
#pragma unroll
                    for (int i = 0; i < NA_LOAD; ++i) if ((i * HS) / NA_LOAD == kk) load_a(i);
#pragma unroll
                    for (int i = 0; i < NB_LOAD; ++i) if ((i * HS) / NB_LOAD == kk) load_b(i);
                } else {
#pragma unroll
                    for (int i = 0; i < NA_LOAD; ++i) if ((i * HS) / NA_LOAD == kk - HS) store_a(cur ^ 1, i);
#pragma unroll
                    for (int i = 0; i < NB_LOAD; ++i) if ((i * HS) / NB_LOAD == kk - HS) store_b(cur ^ 1, i);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][i], bv[cb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    }
    float* out = g.out + (long)blockIdx.z * g.M * g.J;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int jj = j0 + wave_n * 32 * WN + j * 32 + acol;
        if (jj >= g.J) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                if (m < g.M) out[(long)m * g.J + jj] = acc[i][j][r];
            }
    }
}

__global__ void k_reduce_slabs(const float* __restrict__ slabs, float* __restrict__ out, long n, int splits, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = accumulate ? out[i] : 0.f;
        for (int z = 0; z < splits; ++z) s += slabs[(long)z * n + i];
        out[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------
// Weight re-layouts for the data gradient (tiny, once per backward):
//   stride 1:  Wt[ci][(co, a, b)] = W[co][ci][KH-1-a][KW-1-b]
//   stride 2:  Wt[ci][(co, a, b)] = W[co][ci][kh0+2a][kw0+2b]   (taps of one output-parity class)
__global__ void k_weight_relayout(const float* __restrict__ W, float* __restrict__ Wt, int Co, int Ci, int KH, int KW,
                                  int TA, int TB, int kh0, int dkh, int kw0, int dkw) {
    const long n = (long)Ci * Co * TA * TB;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i % TB);
        const int a = (int)((i / TB) % TA);
        const int co = (int)((i / ((long)TB * TA)) % Co);
        const int ci = (int)(i / ((long)TB * TA * Co));
        const int kh = kh0 + dkh * a, kw = kw0 + dkw * b;
        Wt[i] = W[(((long)co * Ci + ci) * KH + kh) * KW + kw];
    }
}

// All weight re-layouts of a training step in one launch: workgroup -> (job, unit) by binary search over the jobs'
// first_block prefix.  For kernels up to 3x3 a unit is a 32 (Cout) x 32 (Cin) tile staged through LDS: the source rows
// W[co][ci0..ci0+31][kh][kw] are contiguous (coalesced reads) and every destination layout has a contiguous run of 32
// along Cin (forward) or Cout (data gradient), so both sides move full 128-byte lines (the per-element gather read with a
// 36-byte stride: 9x the L2 traffic).  Larger kernels (5x5 PoseCNN) keep the per-element path, 1024 elements per unit.
constexpr int RL_TILE = 32;
__host__ __device__ inline bool relayout_tiled(int KH, int KW) { return KH * KW <= 9; }
__host__ __device__ inline long relayout_units(const fd_relayout_job& j) {
    if (relayout_tiled(j.KH, j.KW)) return (long)((j.Co + RL_TILE - 1) / RL_TILE) * ((j.Ci + RL_TILE - 1) / RL_TILE);
    return (j.n + 1023) / 1024;
}

__global__ void __launch_bounds__(256) k_relayout_batch(const fd_relayout_job* __restrict__ jobs, int njobs) {
    __shared__ float tile[RL_TILE][RL_TILE * 9 + 1];
    const long b = blockIdx.x;
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {                                   // last job with first_block <= b
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const fd_relayout_job j = jobs[lo];
    const unsigned u = (unsigned)(b - j.first_block);
    const unsigned TA = (unsigned)j.TA, TB = (unsigned)j.TB, Co = (unsigned)j.Co, Ci = (unsigned)j.Ci;
    if (relayout_tiled(j.KH, j.KW)) {
        const unsigned KK = (unsigned)(j.KH * j.KW), T = TA * TB;
        const unsigned tiles_ci = (Ci + RL_TILE - 1) / RL_TILE;
        const unsigned co0 = (u / tiles_ci) * RL_TILE, ci0 = (u % tiles_ci) * RL_TILE;
        const unsigned nco = Co - co0 < RL_TILE ? Co - co0 : RL_TILE, nci = Ci - ci0 < RL_TILE ? Ci - ci0 : RL_TILE;
        const unsigned row = nci * KK;                   // contiguous source floats per output channel
        for (unsigned i = threadIdx.x; i < nco * row; i += 256) {
            const unsigned r = i / row, q = i - r * row;
            tile[r][q] = j.w[((size_t)(co0 + r) * Ci + ci0) * KK + q];
        }
        __syncthreads();
        if (j.mode == 0) {                               // dst[(co * T + t) * Ci + ci], ci fastest
            for (unsigned i = threadIdx.x; i < nco * T * nci; i += 256) {
                const unsigned c = i % nci, q = i / nci, t = q % T, r = q / T;
                const unsigned a = t / TB, bb = t - a * TB;
                const unsigned tap = (unsigned)(j.kh0 + j.dkh * (int)a) * (unsigned)j.KW + (unsigned)(j.kw0 + j.dkw * (int)bb);
                j.dst[((size_t)(co0 + r) * T + t) * Ci + ci0 + c] = tile[r][c * KK + tap];
            }
        } else if (j.mode == 3) {                        // Winograd U[t][co][ky][ci] (conv_wino.hip), ci fastest
            const size_t n = (size_t)Co * 3 * Ci;
            for (unsigned i = threadIdx.x; i < nco * 3 * nci; i += 256) {
                const unsigned c = i % nci, q = i / nci, ky = q % 3, r = q / 3;
                const float g0 = tile[r][c * 9 + ky * 3], g1 = tile[r][c * 9 + ky * 3 + 1], g2 = tile[r][c * 9 + ky * 3 + 2];
                float* o = j.dst + ((size_t)(co0 + r) * 3 + ky) * Ci + ci0 + c;
                o[0] = g0; o[n] = 0.5f * (g0 + g1 + g2); o[2 * n] = 0.5f * (g0 - g1 + g2); o[3 * n] = g2;
            }
        } else if (j.mode == 4) {                        // Winograd U of the data gradient: [t][ci][ky][co] of the flipped kernel
            const size_t n = (size_t)Ci * 3 * Co;
            for (unsigned i = threadIdx.x; i < nci * 3 * nco; i += 256) {
                const unsigned r = i % nco, q = i / nco, ky = q % 3, c = q / 3;
                const unsigned row = (2 - ky) * 3;
                const float g0 = tile[r][c * 9 + row + 2], g1 = tile[r][c * 9 + row + 1], g2 = tile[r][c * 9 + row];
                float* o = j.dst + ((size_t)(ci0 + c) * 3 + ky) * Co + co0 + r;
                o[0] = g0; o[n] = 0.5f * (g0 + g1 + g2); o[2 * n] = 0.5f * (g0 - g1 + g2); o[3 * n] = g2;
            }
        } else if (j.mode == 5 || j.mode == 6) {         // F(2x2, 3x3): U2[t][co][ri][ci] (5) / [t][ci][ri][co] of the flipped kernel (6)
            const bool dg = j.mode == 6;
            const size_t n = (size_t)Co * 4 * Ci;
            for (unsigned i = threadIdx.x; i < nco * 4 * nci; i += 256) {
                unsigned r, c, ri;
                if (!dg) { c = i % nci; const unsigned q = i / nci; ri = q % 4; r = q / 4; }
                else { r = i % nco; const unsigned q = i / nco; ri = q % 4; c = q / 4; }
                float v[3];
                for (unsigned b = 0; b < 3; ++b) {
                    const unsigned kb = dg ? 2 - b : b;
                    const float g0 = tile[r][c * 9 + (dg ? 6 : 0) + kb], g1 = tile[r][c * 9 + 3 + kb], g2 = tile[r][c * 9 + (dg ? 0 : 6) + kb];
                    v[b] = ri == 0 ? g0 : (ri == 3 ? g2 : (ri == 1 ? 0.5f * (g0 + g1 + g2) : 0.5f * (g0 - g1 + g2)));
                }
                float* o = dg ? j.dst + ((size_t)(ci0 + c) * 4 + ri) * Co + co0 + r : j.dst + ((size_t)(co0 + r) * 4 + ri) * Ci + ci0 + c;
                o[0] = v[0]; o[n] = 0.5f * (v[0] + v[1] + v[2]); o[2 * n] = 0.5f * (v[0] - v[1] + v[2]); o[3 * n] = v[2];
            }
        } else if (j.mode == 11 || j.mode == 12) {       // F(2x2, 3x3): the limb image of U2 (conv_wino.hip: wino_limb_piece); 12: of the flipped kernel, m = ci
            const bool dg = j.mode == 12;
            const unsigned nm = dg ? nci : nco, nk8 = (dg ? nco : nci) / 8;
            const long Mrows = dg ? Ci : Co;
            const int cpt = (int)((dg ? Co : Ci) >> 4);
            uint4* A3 = reinterpret_cast<uint4*>(j.dst);
            for (unsigned i = threadIdx.x; i < nm * 4 * nk8; i += 256) {
                const unsigned k8 = i % nk8, q = i / nk8, ri = q % 4, mm = q / 4;
                float u[4][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned r = dg ? k8 * 8 + e : mm, c = dg ? mm : k8 * 8 + e;      // tile row = co, tile column block = ci
                    float v[3];
                    for (unsigned b = 0; b < 3; ++b) {
                        const unsigned kb = dg ? 2 - b : b;
                        const float g0 = tile[r][c * 9 + (dg ? 6 : 0) + kb], g1 = tile[r][c * 9 + 3 + kb], g2 = tile[r][c * 9 + (dg ? 0 : 6) + kb];
                        v[b] = ri == 0 ? g0 : (ri == 3 ? g2 : (ri == 1 ? 0.5f * (g0 + g1 + g2) : 0.5f * (g0 - g1 + g2)));
                    }
                    u[0][e] = v[0]; u[1][e] = 0.5f * (v[0] + v[1] + v[2]); u[2][e] = 0.5f * (v[0] - v[1] + v[2]); u[3][e] = v[2];
                }
                const long kk = (long)(dg ? co0 : ci0) + k8 * 8, m = (long)(dg ? ci0 : co0) + mm;
                const long pbase = (long)ri * cpt + (kk >> 4);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    uint4 h, md, l;
                    fdlimb::split8(u[t], h, md, l);
                    const long p0 = (((pbase * 4 + t) * 3) * 2 + ((kk >> 3) & 1)) * Mrows + m;
                    A3[p0] = h; A3[p0 + 2 * Mrows] = md; A3[p0 + 4 * Mrows] = l;
                }
            }
        } else if (j.mode == 7 || j.mode == 8) {         // 1x1 weights pre-split into bf16 limbs (conv_limb.h): 7 A[m = co][k = ci], 8 A[m = ci][k = co]
            const bool tr = j.mode == 8;
            const unsigned nm = tr ? nci : nco, nk8 = (tr ? nco : nci) / 8;
            const long Mrows = tr ? Ci : Co;
            uint4* A3 = reinterpret_cast<uint4*>(j.dst);
            for (unsigned i = threadIdx.x; i < nm * nk8; i += 256) {
                const unsigned k8 = i % nk8, mm = i / nk8;
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tr ? tile[k8 * 8 + e][mm] : tile[mm][k8 * 8 + e];
                uint4 h, md, l;
                fdlimb::split8(x, h, md, l);
                const long kk = (tr ? co0 : ci0) + k8 * 8, m = (tr ? ci0 : co0) + mm;
                A3[fdlimb::a3_piece(kk, 0, m, Mrows)] = h;
                A3[fdlimb::a3_piece(kk, 1, m, Mrows)] = md;
                A3[fdlimb::a3_piece(kk, 2, m, Mrows)] = l;
            }
        } else if (j.mode == 9 || j.mode == 10) {        // the taps' matrix [m][(t, c)] pre-split into bf16 limbs: 9 m = co, c = ci (forward), 10 m = ci, c = co (data gradient)
            const bool tr = j.mode == 10;
            const unsigned nm = tr ? nci : nco, nk8 = (tr ? nco : nci) / 8;
            const long Mrows = tr ? Ci : Co, Cr = tr ? Co : Ci;
            uint4* A3 = reinterpret_cast<uint4*>(j.dst);
            for (unsigned i = threadIdx.x; i < nm * T * nk8; i += 256) {
                const unsigned k8 = i % nk8, q = i / nk8, t = q % T, mm = q / T;
                const unsigned a = t / TB, bb = t - a * TB;
                const unsigned tap = (unsigned)(j.kh0 + j.dkh * (int)a) * (unsigned)j.KW + (unsigned)(j.kw0 + j.dkw * (int)bb);
                float x[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) x[e] = tr ? tile[k8 * 8 + e][mm * KK + tap] : tile[mm][(k8 * 8 + e) * KK + tap];
                uint4 h, md, l;
                fdlimb::split8(x, h, md, l);
                const long kk = (long)t * Cr + (tr ? co0 : ci0) + k8 * 8, m = (tr ? ci0 : co0) + mm;
                A3[fdlimb::a3_piece(kk, 0, m, Mrows)] = h;
                A3[fdlimb::a3_piece(kk, 1, m, Mrows)] = md;
                A3[fdlimb::a3_piece(kk, 2, m, Mrows)] = l;
            }
        } else if (j.mode == 1) {                        // dst[(ci * T + t) * Co + co], co fastest
            for (unsigned i = threadIdx.x; i < nci * T * nco; i += 256) {
                const unsigned r = i % nco, q = i / nco, t = q % T, c = q / T;
                const unsigned a = t / TB, bb = t - a * TB;
                const unsigned tap = (unsigned)(j.kh0 + j.dkh * (int)a) * (unsigned)j.KW + (unsigned)(j.kw0 + j.dkw * (int)bb);
                j.dst[((size_t)(ci0 + c) * T + t) * Co + co0 + r] = tile[r][c * KK + tap];
            }
        } else {                                         // dst[((ci * Co + co) * TA + a) * TB + b], taps fastest
            for (unsigned i = threadIdx.x; i < nci * nco * T; i += 256) {
                const unsigned t = i % T, q = i / T, r = q % nco, c = q / nco;
                const unsigned a = t / TB, bb = t - a * TB;
                const unsigned tap = (unsigned)(j.kh0 + j.dkh * (int)a) * (unsigned)j.KW + (unsigned)(j.kw0 + j.dkw * (int)bb);
                j.dst[((size_t)(ci0 + c) * Co + co0 + r) * T + t] = tile[r][c * KK + tap];
            }
        }
        return;
    }
    const unsigned base = u * 1024u;                     // a job has < 2^31 elements: 32-bit index math
    const unsigned n = (unsigned)j.n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned i = base + e * 256u + threadIdx.x;
        if (i >= n) return;
        unsigned co, ci, a, bb;
        if (j.mode == 2) {                              // [ci][co][a][b]
            unsigned q = i / TB; bb = i - q * TB;
            unsigned q2 = q / TA; a = q - q2 * TA;
            ci = q2 / Co; co = q2 - ci * Co;
        } else {                                        // [m][a][b][c]: mode 0 m = co, c = ci; mode 1 m = ci, c = co
            const unsigned Cr = j.mode ? Co : Ci;
            unsigned q = i / Cr; const unsigned c = i - q * Cr;
            unsigned q2 = q / TB; bb = q - q2 * TB;
            const unsigned m = q2 / TA; a = q2 - m * TA;
            co = j.mode ? c : m; ci = j.mode ? m : c;
        }
        j.dst[i] = j.w[((co * Ci + ci) * (unsigned)j.KH + (unsigned)(j.kh0 + j.dkh * (int)a)) * (unsigned)j.KW +
                       (unsigned)(j.kw0 + j.dkw * (int)bb)];
    }
}

// Reflect-padded data gradient without the padded-grid round trip: the interior of the padded grid IS the zero-padded data gradient
// (written straight to gx by the convolution kernel), only the one-pixel ring needs the adjoint of ReflectionPad2d(1).  The ring
// arrives as four strips per (image, channel) - top [W+2], bottom [W+2], left [H], right [H] (padded rows 1 .. H) - and folds as
//   gx[1][x] += top[x+1], gx[H-2][x] += bottom[x+1], gx[y][1] += left[y], gx[y][W-2] += right[y],
// the corners top[0] / top[W+1] / bottom[0] / bottom[W+1] going to (1,1) / (1,W-2) / (H-2,1) / (H-2,W-2).  One thread per target pixel
// (rows 1 / H-2, columns 1 / W-2) sums every strip value that maps to it in a fixed order: deterministic, also when H - 2 == 1.
__global__ void __launch_bounds__(256) k_reflect_ring_fold(const float* __restrict__ ring, float* __restrict__ gx, long planes, int H,
                                                           int W) {
    const int rows2 = (H - 2 != 1) ? 2 : 1, cols2 = (W - 2 != 1) ? 2 : 1;     // distinct target rows / columns
    const int nrow_t = rows2 * W, ncol_t = (H - rows2) * cols2, nt = nrow_t + ncol_t;
    const long ring_plane = 2L * (W + 2) + 2L * H;
    const long total = planes * nt;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long pl = i / nt;
        const int t = (int)(i - pl * nt);
        int y, x;
        if (t < nrow_t) { const int r = t / W; x = t - r * W; y = r == 0 ? 1 : H - 2; }
        else {
            const int q = t - nrow_t, k = q / (H - rows2), j = q - k * (H - rows2);
            x = k == 0 ? 1 : W - 2;
            const int r_lo = (rows2 == 2 && H - 2 < 1) ? H - 2 : 1, r_hi = (rows2 == 2 && H - 2 < 1) ? 1 : H - 2;   // sorted target rows
            y = j;
            if (y >= r_lo) ++y;
            if (rows2 == 2 && y >= r_hi) ++y;
        }
        const float* top = ring + pl * ring_plane;
        const float* bot = top + (W + 2);
        const float* lef = bot + (W + 2);
        const float* rig = lef + H;
        // padded rows {0 if y == 1, H+1 if y == H-2} x padded columns {x+1, 0 if x == 1, W+1 if x == W-2}; padded row y+1 x
        // padded columns {0 if x == 1, W+1 if x == W-2}
        float s = 0.f;
        if (y == 1) { s += top[x + 1]; if (x == 1) s += top[0]; if (x == W - 2) s += top[W + 1]; }
        if (y == H - 2) { s += bot[x + 1]; if (x == 1) s += bot[0]; if (x == W - 2) s += bot[W + 1]; }
        if (x == 1) s += lef[y];
        if (x == W - 2) s += rig[y];
        gx[pl * (long)H * W + (long)y * W + x] += s;
    }
}

// planes of [H][W] -> planes of [H+2][W+2] with a border of zeros (the padded-grid data gradient on the Winograd kernel, below)
__global__ void __launch_bounds__(256) k_zero_border_copy(const float* __restrict__ src, float* __restrict__ dst, long planes, int H, int W) {
    const int Wp = W + 2, np = (H + 2) * Wp;
    for (long pl = blockIdx.y; pl < planes; pl += gridDim.y)
        for (int r = blockIdx.x * 256 + threadIdx.x; r < np; r += gridDim.x * 256) {
            const int y = r / Wp - 1, x = r - (y + 1) * Wp - 1;
            dst[pl * np + r] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? src[pl * (long)H * W + y * W + x] : 0.f;
        }
}

// Adjoint of ReflectionPad2d(1): fold the gradient on the padded grid [H+2][W+2] back onto [H][W].
__global__ void k_reflect_fold(const float* __restrict__ gp, float* __restrict__ gx, long planes, int H, int W) {
    const int Wp = W + 2;
    for (long pl = blockIdx.y; pl < planes; pl += gridDim.y)
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < H * W; r += gridDim.x * blockDim.x) {
        const int y = r / W, x = r - y * W;
        const long i = pl * H * W + r;
        const float* g = gp + pl * (long)(H + 2) * Wp;
        // padded rows that map to y: y+1 always; 0 if y == 1; H+1 if y == H-2
        int ys[3], nys = 0, xs[3], nxs = 0;
        ys[nys++] = y + 1; if (y == 1) ys[nys++] = 0; if (y == H - 2) ys[nys++] = H + 1;
        xs[nxs++] = x + 1; if (x == 1) xs[nxs++] = 0; if (x == W - 2) xs[nxs++] = W + 1;
        float s = 0.f;
        for (int a = 0; a < nys; ++a)
            for (int b = 0; b < nxs; ++b) s += g[ys[a] * Wp + xs[b]];
        gx[i] = s;
    }
}

// per-channel sum over (n, y, x) — bias gradient.  Two deterministic stages: (channel, slice) partials, then a
// fixed-order sum over the slices.
constexpr int CS_SPLITS = 32;
// Round 5: one 64-bit division per ELEMENT and scalar loads made this pass run at 0.6 TB/s (150 us for the 94 MB of upconv(0, 1)'s
// gradient); now slice s of every image's plane, float4 loads four at a time, no index arithmetic in the loop.  The summation order
// changed with it (per thread: images in order, its quads in order; then the fixed block tree) - still deterministic.
template <bool VEC>
__global__ void __launch_bounds__(256) k_channel_sum_part(const float* __restrict__ x, float* __restrict__ part, int Nb,
                                                          int C, long plane) {
    __shared__ float red[4];
    const int c = blockIdx.x, s = blockIdx.y;
    float v[1] = {0.f};
    if (VEC) {
        const long q = plane >> 2, per = (q + CS_SPLITS - 1) / CS_SPLITS;
        const long lo = (long)s * per, hi = lo + per < q ? lo + per : q;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int n = 0; n < Nb; ++n) {
            const float4* p = reinterpret_cast<const float4*>(x + ((long)n * C + c) * plane);
            long i = lo + threadIdx.x;
            for (; i + 768 < hi; i += 1024) {                       // four independent loads in flight
                const float4 u0 = p[i], u1 = p[i + 256], u2 = p[i + 512], u3 = p[i + 768];
                a0 += (u0.x + u0.y) + (u0.z + u0.w); a1 += (u1.x + u1.y) + (u1.z + u1.w);
                a2 += (u2.x + u2.y) + (u2.z + u2.w); a3 += (u3.x + u3.y) + (u3.z + u3.w);
            }
            for (; i < hi; i += 256) { const float4 u = p[i]; a0 += (u.x + u.y) + (u.z + u.w); }
        }
        v[0] = (a0 + a1) + (a2 + a3);
    } else {
        const long per = (plane + CS_SPLITS - 1) / CS_SPLITS;
        const long lo = (long)s * per, hi = lo + per < plane ? lo + per : plane;
        for (int n = 0; n < Nb; ++n) {
            const float* p = x + ((long)n * C + c) * plane;
            for (long i = lo + threadIdx.x; i < hi; i += 256) v[0] += p[i];
        }
    }
    const float sum = fd_block_sum_n<1, 4>(v, red);
    if (threadIdx.x == 0) part[(long)c * CS_SPLITS + s] = sum;
}
__global__ void k_channel_sum_fin(const float* __restrict__ part, float* __restrict__ out, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = accumulate ? out[c] : 0.f;
    for (int i = 0; i < CS_SPLITS; ++i) s += part[(long)c * CS_SPLITS + i];
    out[c] = s;
}

// ------------------------------------------------------------------------------------------------
template <int TA, int TB>
int launch_gemm(const GemmArgs& g, hipStream_t st) {
    const long Np = (long)g.Nb * g.NY * g.NX;
    GemmArgs a = g;
    auto go = [&](auto kern, int BM, int BN, int nt) {
        const int gx = fd_cdiv(Np, BN), gy = fd_cdiv(g.M, BM);
        a.xcd_swizzle = (gx % 8 == 0 && gx >= 16) ? 1 : 0;
        hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(nt), 0, st, a);
    };
    // pick the largest tile that still yields >= ~2 workgroups per CU
    auto blocks = [&](int BM, int BN) { return (long)fd_cdiv(Np, BN) * fd_cdiv(g.M, BM); };
    constexpr bool CAN_NORM = (TA == 7 && TB == 7);
    if (g.in_norm && !CAN_NORM) { fd_set_error("conv: in_norm is only built for the 7x7 stem"); return -1; }
    if (CAN_NORM && g.in_norm) {
        if (blocks(64, 128) >= 512) go(k_gather_gemm<TA, TB, 2, 2, 1, 2, CAN_NORM>, 64, 128, 256);
        else go(k_gather_gemm<TA, TB, 2, 2, 1, 1, CAN_NORM>, 64, 64, 256);
        return 0;
    }
    if (g.M <= 32) {
        go(k_gather_gemm<TA, TB, 1, 4, 1, 1, false>, 32, 128, 256);
    } else if (g.M >= 128 && blocks(128, 128) >= 512) {
        go(k_gather_gemm<TA, TB, 2, 2, 2, 2, false>, 128, 128, 256);
    } else if (blocks(64, 128) >= 512) {
        go(k_gather_gemm<TA, TB, 2, 2, 1, 2, false>, 64, 128, 256);
    } else {
        go(k_gather_gemm<TA, TB, 2, 2, 1, 1, false>, 64, 64, 256);
    }
    return 0;
}

int dispatch_gemm(int TA, int TB, const GemmArgs& g, hipStream_t st) {
    if (TA == 1 && TB == 1) return launch_gemm<1, 1>(g, st);
    if (TA == 3 && TB == 3) return launch_gemm<3, 3>(g, st);
    if (TA == 7 && TB == 7) return launch_gemm<7, 7>(g, st);
    if (TA == 5 && TB == 5) return launch_gemm<5, 5>(g, st);
    if (TA == 1 && TB == 2) return launch_gemm<1, 2>(g, st);
    if (TA == 2 && TB == 1) return launch_gemm<2, 1>(g, st);
    if (TA == 2 && TB == 2) return launch_gemm<2, 2>(g, st);
    if (TA == 3 && TB == 4) return launch_gemm<3, 4>(g, st);
    if (TA == 4 && TB == 3) return launch_gemm<4, 3>(g, st);
    if (TA == 4 && TB == 4) return launch_gemm<4, 4>(g, st);
    if (TA == 2 && TB == 3) return launch_gemm<2, 3>(g, st);
    if (TA == 3 && TB == 2) return launch_gemm<3, 2>(g, st);
    fd_set_error("conv: unsupported tap shape %dx%d", TA, TB);
    return -1;
}

template <int TA, int TB>
int launch_wgrad(const WgradArgs& g, int splits, hipStream_t st) {
    constexpr bool CAN_REFL = (TA == 3 && TB == 3), CAN_NORM = (TA == 7 && TB == 7);
    if (g.pad_mode == 1 && !CAN_REFL) { fd_set_error("conv wgrad: reflect padding is only built for 3x3"); return -1; }
    if (g.in_norm && !CAN_NORM) { fd_set_error("conv wgrad: in_norm is only built for the 7x7 stem"); return -1; }
    auto go = [&](auto kern, int BM, int BN) {
        dim3 grid(fd_cdiv(g.J, BN), fd_cdiv(g.M, BM), splits);
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, st, g);
    };
    const bool refl = CAN_REFL && g.pad_mode == 1, norm = CAN_NORM && g.in_norm;
    if (g.J <= 64) {
        if (refl) go(k_wgrad<TA, TB, 2, 2, 1, 1, CAN_REFL, false>, 64, 64);
        else if (norm) go(k_wgrad<TA, TB, 2, 2, 1, 1, false, CAN_NORM>, 64, 64);
        else go(k_wgrad<TA, TB, 2, 2, 1, 1, false, false>, 64, 64);
    } else if (g.M <= 32) {
        if (refl) go(k_wgrad<TA, TB, 1, 4, 1, 1, CAN_REFL, false>, 32, 128);
        else if (norm) go(k_wgrad<TA, TB, 1, 4, 1, 1, false, CAN_NORM>, 32, 128);
        else go(k_wgrad<TA, TB, 1, 4, 1, 1, false, false>, 32, 128);
    } else {
        if (refl) go(k_wgrad<TA, TB, 2, 2, 1, 2, CAN_REFL, false>, 64, 128);
        else if (norm) go(k_wgrad<TA, TB, 2, 2, 1, 2, false, CAN_NORM>, 64, 128);
        else go(k_wgrad<TA, TB, 2, 2, 1, 2, false, false>, 64, 128);
    }
    return 0;
}

int dispatch_wgrad(int TA, int TB, const WgradArgs& g, int splits, hipStream_t st) {
    if (TA == 1 && TB == 1) return launch_wgrad<1, 1>(g, splits, st);
    if (TA == 3 && TB == 3) return launch_wgrad<3, 3>(g, splits, st);
    if (TA == 7 && TB == 7) return launch_wgrad<7, 7>(g, splits, st);
    if (TA == 5 && TB == 5) return launch_wgrad<5, 5>(g, splits, st);
    fd_set_error("conv wgrad: unsupported kernel %dx%d", TA, TB);
    return -1;
}

struct ConvShape {
    int Ho, Wo;
};
bool conv_out_shape(const fd_conv_desc* d, ConvShape& s) {
    s.Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1;
    s.Wo = (d->W + 2 * d->pad - d->KW) / d->stride + 1;
    return s.Ho > 0 && s.Wo > 0;
}
int check_desc(const fd_conv_desc* d, const char* who) {
    FD_REQUIRE(d, "%s: desc is NULL", who);
    FD_REQUIRE(d->N > 0 && d->Cin > 0 && d->Cout > 0 && d->H > 0 && d->W > 0, "%s: bad sizes", who);
    FD_REQUIRE(d->KH == d->KW && (d->KH == 1 || d->KH == 3 || d->KH == 5 || d->KH == 7), "%s: kernel %dx%d unsupported", who,
               d->KH, d->KW);
    FD_REQUIRE(d->stride == 1 || d->stride == 2, "%s: stride %d unsupported", who, d->stride);
    FD_REQUIRE(d->pad >= 0 && d->pad <= d->KH / 2 + 1, "%s: pad %d unsupported", who, d->pad);
    FD_REQUIRE(d->pad_mode == 0 || (d->pad_mode == 1 && d->stride == 1 && d->pad == 1 && d->H >= 2 && d->W >= 2),
               "%s: reflect padding needs stride 1, pad 1", who);
    FD_REQUIRE(d->act >= 0 && d->act <= 4, "%s: bad activation", who);
    return 0;
}
inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

namespace {
inline long align4(long n) { return (n + 3) / 4 * 4; }
inline bool fast_fwd_ok(const fd_conv_desc* d) { return d->Cin % 16 == 0 && !d->in_norm; }
inline bool fast_dgrad_ok(const fd_conv_desc* d) { return d->Cout % 16 == 0; }
// the stride-2 layers on the split-precision implicit GEMM (conv_limb.hip: k_conv_limb), forward and data gradient
// (3x3 kernels always; the 1x1 stride-2 downsample layers only where the launch fills the chip without split-K - limb_conv_1x1_worth)
inline bool limb_conv_fwd_ok(const fd_conv_desc* d) {
    if (!(d->stride == 2 && !d->in_norm && limb_conv_problem_ok(d->Cout, d->Cin, d->pad_mode, d->act))) return false;
    if (d->KH * d->KW > 1) return true;
    ConvShape s;
    return d->pad == 0 && conv_out_shape(d, s) && limb_conv_1x1_worth(d->Cout, (long)d->N * s.Ho * s.Wo);
}
inline bool limb_conv_dgrad_ok(const fd_conv_desc* d) {
    if (!(d->stride == 2 && d->pad_mode == 0 && limb_conv_problem_ok(d->Cin, d->Cout, 0, 0))) return false;
    if (d->KH * d->KW > 1) return true;
    ConvShape s;
    return d->pad == 0 && conv_out_shape(d, s) && limb_conv_1x1_worth(d->Cin, (long)d->N * s.Ho * s.Wo);
}
inline bool limb_conv_wgrad_ok(const fd_conv_desc* d, const ConvShape& s) {
    if (!(d->stride == 2 && d->pad_mode == 0 && !d->in_norm && limb_wgrad_s2_shape_ok(d->Cout, d->Cin, d->H, d->W, s.Ho, s.Wo))) return false;
    if (d->KH == 3 && d->KW == 3 && d->pad == 1) return true;
    return d->KH == 1 && d->KW == 1 && d->pad == 0 && d->Cin >= 256;      // (K = pixels here: the short dimension is Cin x Cout tiles)
}
inline bool fast_wgrad_ok(const fd_conv_desc* d) { return d->Cin % 16 == 0 && d->Cin >= 64 && !d->in_norm; }   // narrow layers: a (tap, channel) tile would be mostly padding

// fd_tuning.log: one stderr line per convolution call (which kernel family it was routed to) - a tuning aid
void conv_log(const char* what, const char* path, const fd_conv_desc* d) {
    if (fd_tun().log) fprintf(stderr, "FDCONV %s %s N=%d Cin=%d H=%d W=%d Cout=%d K=%d s=%d pad_mode=%d\n", what, path, d->N, d->Cin, d->H, d->W, d->Cout, d->KH,
                              d->stride, d->pad_mode);
}
// 1-D Winograd F(2,3) path (conv_wino.hip): 3x3 stride-1 pad-1 convs with >= 64 output channels (its tile is 64 channels tall).
// layer1's 64x64 weight gradient (3 tiles x 256 pixel-splits) is 10 % slower than the direct kernel when run alone and still the
// better choice inside the step (449.6 vs 442 images/s): what the step is short of is MFMA cycles, not launch latency
inline bool wino_use_wgrad(const fd_conv_desc* d) { return fd_tun().wino_wgrad != 0 && wino_wgrad_ok(d); }
// fd_tuning.wino_fwd = 0: forward and data gradient stay on the direct kernels (A/B runs)
bool wino_fwd_enabled() { return fd_tun().wino_fwd != 0; }
inline bool wino_use_fwd(const fd_conv_desc* d) { return wino_fwd_enabled() && wino_fwd_ok(d) && d->Cout >= fd_tun().wino_min_cout; }
// the data gradient of a zero-padded 3x3 stride-1 conv is the same kind of conv over dY (channels swapped, kernel flipped)
inline bool wino_dgrad_desc(const fd_conv_desc* d, fd_conv_desc& g) {
    if (!(wino_fwd_enabled() && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->pad_mode == 0)) return false;
    g = *d;
    g.Cin = d->Cout; g.Cout = d->Cin; g.act = 0; g.in_norm = 0;
    return wino_fwd_ok(&g) && g.Cout >= fd_tun().wino_min_cout;
}

// The interior of a REFLECT-padded 3x3 data gradient (the padded grid's cells that are real pixels) is the zero-padded data
// gradient: a plain 3x3 convolution over dY with the transposed, flipped kernel - Winograd-eligible like the trunk's.  With the
// padded grid's one-pixel ring as four thin problems on the implicit-GEMM kernel + k_reflect_ring_fold (round 3), the decoder's
// wide blocks (upconv(2..4, *): 64 .. 512 channels) leave the direct kernel's padded-grid pass (46 - 60 TFLOP/s on these shapes)
// and its full fold pass.  `g`: the convolution the interior computes.  fd_tuning.reflect_wino = 0 switches it off; needs reflect_ring != 0.
bool refl_wino_interior(const fd_conv_desc* d, fd_conv_desc& g) {
    if (!(wino_fwd_enabled() && d->pad_mode == 1 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->H >= 2 && d->W >= 2)) return false;
    const fd_tuning& t = fd_tun();
    if (!t.reflect_wino || !t.reflect_ring) return false;
    if ((long)d->H * d->W < (long)t.reflect_wino_min_pixels) return false;
    if (!fast_dgrad_ok(d)) return false;                     // the ring runs on the implicit-GEMM kernel
    g = *d;
    g.Cin = d->Cout; g.Cout = d->Cin; g.pad_mode = 0; g.act = 0; g.in_norm = 0;
    return wino_fwd_ok(&g) && g.Cout >= fd_tun().wino_min_cout;
}

// ... and on SMALL planes (below 4 096 pixels: upconv(2..4, *) at 6x20 .. 24x80) the ring's four thin problems -
// few pixels against up to 512 output channels, 32 - 96 workgroups with 72 chunks each and no split-K - cost more than the interior
// (189 us against 60 us for upconv(4,1)).  There the whole padded-grid gradient is ONE Winograd convolution over dY embedded in a
// border of zeros ((H+2) x (W+2): 6 - 47 % more pixels), followed by the fold pass.  `gp`: that convolution.
bool refl_wino_padded(const fd_conv_desc* d, fd_conv_desc& gp) {
    fd_conv_desc gz;
    if (!refl_wino_interior(d, gz)) return false;
    const fd_tuning& t = fd_tun();
    long thr = t.reflect_wino_padded_max;                    // measured in the step: 4 096 (planes up to 24x80) 20.35 - 20.43 ms, 16 384 20.44 - 20.51, 65 536 20.54 - 20.59; 0: never
    if (t.reflect_ring > 1 && t.reflect_ring < thr) thr = t.reflect_ring;    // (the tests' "ring from n pixels on")
    if ((long)d->H * d->W >= thr) return false;              // larger planes: interior + ring
    gp = gz;
    gp.H = d->H + 2; gp.W = d->W + 2;
    return wino_fwd_ok(&gp);
}

void fill_fwd_args(const fd_conv_desc* d, const ConvShape& s, FastGemmArgs& f) {
    f = FastGemmArgs{};
    f.M = d->Cout; f.C = d->Cin; f.T = d->KH * d->KW; f.TB = d->KW; f.K = f.T * f.C;
    f.Nb = d->N; f.Hi = d->H; f.Wi = d->W; f.NY = s.Ho; f.NX = s.Wo;
    f.sy = d->stride; f.oy = -d->pad; f.da = 1; f.sx = d->stride; f.ox = -d->pad; f.db = 1;
    f.pad_mode = d->pad_mode;
    f.out_cs = (long)s.Ho * s.Wo; f.out_ns = f.out_cs * d->Cout; f.out_total = f.out_ns * d->N;
    f.slab_stride = f.out_total;
    f.out_w = s.Wo; f.osy = 1; f.ooy = 0; f.osx = 1; f.oox = 0;
    f.act = d->act;
}
}  // namespace

extern "C" long fd_conv2d_fwd_wt_floats(const fd_conv_desc* d) {
    if (!d || c1_shape_ok(d) || !fast_fwd_ok(d)) return 0;
    if (n16_shape_ok(d, d->Cout, d->Cin)) return 0;              // reads the weights as they are
    if (limb_fwd_ok(d)) return align4(limb_wt_floats(d->Cout, d->Cin));
    if (wino_use_fwd(d)) return align4(wino_wt_floats(d));
    if (limb_conv_fwd_ok(d)) return align4(limb_wt_floats(d->Cout, (long)d->KH * d->KW * d->Cin));
    return align4((long)d->Cout * d->Cin * d->KH * d->KW);
}

extern "C" long fd_conv2d_fwd_ws_floats(const fd_conv_desc* d) {
    if (!d) return 0;
    ConvShape s;
    if (!conv_out_shape(d, s) || c1_shape_ok(d) || !fast_fwd_ok(d)) return 0;
    if (n16_shape_ok(d, d->Cout, d->Cin)) return 0;
    if (limb_fwd_ok(d)) return limb_gemm_ws_floats(d->Cout, d->Cin, d->N, d->H * d->W);
    if (wino_use_fwd(d)) return wino_ws_floats(d);
    FastGemmArgs f;
    fill_fwd_args(d, s, f);
    if (limb_conv_fwd_ok(d)) return limb_conv_ws_floats(f);
    return fast_splitk_slab_floats(f, nullptr);
}

namespace {
int conv2d_fwd_impl(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt, int wt_ready,
                    float* ws, float* stat_part, void* stream);
}
extern "C" int fd_conv2d_fwd(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                             int wt_ready, float* ws, void* stream) {
    return conv2d_fwd_impl(d, x, w, bias, y, wt, wt_ready, ws, nullptr, stream);
}
extern "C" int fd_conv2d_fwd_bn_ok(const fd_conv_desc* d, int groups) {
    if (!d || check_desc(d, "fd_conv2d_fwd_bn_ok")) return 0;
    if (!(fast_fwd_ok(d) && wino_use_fwd(d) && wino_fwd_slab_route(d)) || d->act != 0 || c1_shape_ok(d) || n16_shape_ok(d, d->Cout, d->Cin)) return 0;
    return bn_small_slabs_ok(d->N, d->Cout, d->H, d->W, groups) ? 1 : 0;
}
extern "C" int fd_conv2d_fwd_bn(const fd_conv_desc* d, const float* x, const float* w, float* y, float* wt, int wt_ready, float* ws,
                                const float* bn_weight, const float* bn_bias, const float* residual, float* out, float* running_mean,
                                float* running_var, float* save_mean, float* save_invstd, int groups, float eps, float momentum, int relu,
                                void* stream) {
    FD_REQUIRE(fd_conv2d_fwd_bn_ok(d, groups), "fd_conv2d_fwd_bn: not a slab-route 3x3 convolution followed by a small-plane BatchNorm (fd_conv2d_fwd_bn_ok == 0)");
    FD_REQUIRE(x && w && y && wt && ws && out && save_mean && save_invstd, "fd_conv2d_fwd_bn: NULL argument");
    FD_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "fd_conv2d_fwd_bn: running stats must come in pairs");
    hipStream_t st = (hipStream_t)stream;
    conv_log("fwd", "wino + bn", d);
    if (!wt_ready)
        if (int rc = wino_weight_launch(d, w, wt, 0, st)) return rc;
    BnAfterConv bn = {bn_weight, bn_bias, residual, out, running_mean, running_var, save_mean, save_invstd, groups, eps, momentum, relu};
    return wino_conv_launch(d, x, wt, nullptr, y, ws, st, nullptr, nullptr, &bn);
}
extern "C" long fd_conv2d_fwd_stat_slots(const fd_conv_desc* d) {
    if (!d || check_desc(d, "fd_conv2d_fwd_stat_slots")) return 0;
    return (fast_fwd_ok(d) && wino_use_fwd(d)) ? wino_stat_slots(d) : 0;
}
extern "C" int fd_conv2d_fwd_stats(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                                   int wt_ready, float* ws, float* stat_part, void* stream) {
    FD_REQUIRE(stat_part, "fd_conv2d_fwd_stats: stat_part is NULL");
    FD_REQUIRE(fd_conv2d_fwd_stat_slots(d) > 0, "fd_conv2d_fwd_stats: this convolution has no statistics epilogue (fd_conv2d_fwd_stat_slots == 0)");
    return conv2d_fwd_impl(d, x, w, bias, y, wt, wt_ready, ws, stat_part, stream);
}
namespace {
int conv2d_fwd_impl(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt, int wt_ready,
                    float* ws, float* stat_part, void* stream) {
    if (int rc = check_desc(d, "fd_conv2d_fwd")) return rc;
    FD_REQUIRE(x && w && y, "fd_conv2d_fwd: NULL tensor");
    ConvShape s;
    FD_REQUIRE(conv_out_shape(d, s), "fd_conv2d_fwd: empty output");
    FD_REQUIRE((long)d->N * d->Cin * d->H * d->W < (1L << 29) && (long)d->N * d->Cout * s.Ho * s.Wo < (1L << 29),
               "fd_conv2d_fwd: tensor too large for 32-bit byte offsets (2 GiB per tensor)");
    hipStream_t st = (hipStream_t)stream;
    if (c1_shape_ok(d)) {
        FD_REQUIRE(!stat_part, "fd_conv2d_fwd_stats: no statistics epilogue for this shape");
        conv_log("fwd", "c1 stencil", d);
        return c1_fwd_launch(d, x, w, bias, y, st);
    }
    if (fast_fwd_ok(d) && n16_shape_ok(d, d->Cout, d->Cin)) {
        FD_REQUIRE(!stat_part, "fd_conv2d_fwd_stats: no statistics epilogue for this shape");
        conv_log("fwd", "n16", d);
        return n16_launch(d, d->Cout, d->Cin, x, w, bias, y, 0, d->pad_mode, d->act, st);
    }
    if (!bias && stem7_fwd_ok(d)) {
        FD_REQUIRE(!stat_part, "fd_conv2d_fwd_stats: no statistics epilogue for this shape");
        conv_log("fwd", "stem7", d);
        return stem7_fwd_launch(d, x, w, bias, y, st);
    }
    if (fast_fwd_ok(d)) {
        FD_REQUIRE(wt, "fd_conv2d_fwd: weight-layout buffer required (fd_conv2d_fwd_wt_floats)");
        if (limb_fwd_ok(d)) {
            FD_REQUIRE(!stat_part, "fd_conv2d_fwd_stats: no statistics epilogue for this shape");
            conv_log("fwd", "limb 1x1", d);
            if (!wt_ready)
                if (int rc = limb_weight_split_launch(w, wt, d->Cout, d->Cin, 0, st)) return rc;
            return limb_gemm_launch(wt, x, y, bias, nullptr, ws, d->Cout, d->Cin, d->N, d->H * d->W, d->act, st);
        }
        if (wino_use_fwd(d)) {
            conv_log("fwd", "wino", d);
            if (!wt_ready)
                if (int rc = wino_weight_launch(d, w, wt, 0, st)) return rc;
            return wino_conv_launch(d, x, wt, bias, y, ws, st, nullptr, stat_part);
        }
        FastGemmArgs f;
        fill_fwd_args(d, s, f);
        if (limb_conv_fwd_ok(d)) {
            FD_REQUIRE(!stat_part, "fd_conv2d_fwd_stats: no statistics epilogue for this shape");
            conv_log("fwd", "limb direct", d);
            if (!wt_ready)
                if (int rc = limb_conv_weight_split_launch(w, wt, d->Cout, d->Cin, d->KH, d->KW, d->KH, d->KW, 0, 1, 0, 1, 0, st)) return rc;
            f.A = wt; f.X = x; f.Y = y; f.bias = bias;
            f.slabs = ws;
            return limb_conv_launch(f, st);
        }
        conv_log("fwd", "direct", d);
        if (!wt_ready)
            if (int rc = fast_weight_relayout(w, wt, d->Cout, d->Cin, d->KH, d->KW, d->KH, d->KW, 0, 1, 0, 1, 0, st)) return rc;
        f.A = wt; f.X = x; f.Y = y; f.bias = bias;
        f.slabs = ws;
        return fast_gemm_launch(f, st);
    }
    GemmArgs g = {};
    g.A = w; g.X = x; g.Y = y; g.bias = bias;
    g.M = d->Cout; g.K = d->Cin * d->KH * d->KW;
    g.Nb = d->N; g.C = d->Cin; g.Hi = d->H; g.Wi = d->W;
    g.NY = s.Ho; g.NX = s.Wo;
    g.sy = d->stride; g.oy = -d->pad; g.da = 1; g.sx = d->stride; g.ox = -d->pad; g.db = 1;
    g.pad_mode = d->pad_mode;
    g.out_ns = (long)d->Cout * s.Ho * s.Wo; g.out_cs = (long)s.Ho * s.Wo;
    g.out_w = s.Wo; g.osy = 1; g.ooy = 0; g.osx = 1; g.oox = 0;
    g.act = d->act; g.in_norm = d->in_norm;
    if (int rc = dispatch_gemm(d->KH, d->KW, g, st)) return rc;
    FD_LAUNCH_CHECK("fd_conv2d_fwd");
    return 0;
}
}  // namespace

extern "C" long fd_conv2d_bwd_data_wt_floats(const fd_conv_desc* d) {
    if (!d || c1_shape_ok(d)) return 0;
    if (limb_dgrad_ok(d)) return align4(limb_wt_floats(d->Cin, d->Cout));
    fd_conv_desc g;
    if (wino_dgrad_desc(d, g)) return align4(wino_wt_floats(&g));
    if (refl_wino_padded(d, g)) return align4(wino_wt_floats(&g));
    if (refl_wino_interior(d, g))                            // [layout of the ring's implicit GEMM | U of the interior's Winograd kernel]
        return align4((long)d->Cin * d->Cout * d->KH * d->KW) + align4(wino_wt_floats(&g));
    return (d->stride == 1 ? 1 : 4) * align4((long)d->Cin * d->Cout * d->KH * d->KW);
}

extern "C" long fd_conv2d_bwd_data_ws_floats(const fd_conv_desc* d) {
    if (!d) return 0;
    ConvShape s;
    if (!conv_out_shape(d, s) || c1_shape_ok(d)) return 0;
    if (limb_dgrad_ok(d)) return limb_gemm_ws_floats(d->Cin, d->Cout, d->N, d->H * d->W);
    {
        fd_conv_desc g;
        if (wino_dgrad_desc(d, g)) return wino_ws_floats(&g);
    }
    const long wt = 0;
    const long padded = d->pad_mode == 1 ? align4((long)d->N * d->Cin * (d->H + 2) * (d->W + 2)) : 0;
    long slabs = 0;
    if (fast_dgrad_ok(d) && d->stride == 1) {          // split-K only on the stride-1 path
        FastGemmArgs f = {};
        f.M = d->Cin; f.C = d->Cout; f.T = d->KH * d->KW; f.Nb = d->N;
        f.NY = d->pad_mode == 1 ? d->H + 2 : d->H; f.NX = d->pad_mode == 1 ? d->W + 2 : d->W;
        f.osy = 1; f.osx = 1;
        f.out_total = (long)d->N * d->Cin * f.NY * f.NX;
        slabs = fast_splitk_slab_floats(f, nullptr);
        if (d->pad_mode == 1) {                        // the interior-plus-ring path runs the H x W problem: its own split count
            f.NY = d->H; f.NX = d->W;
            f.out_total = (long)d->N * d->Cin * f.NY * f.NX;
            const long s2 = fast_splitk_slab_floats(f, nullptr);
            slabs = s2 > slabs ? s2 : slabs;
            fd_conv_desc gz;
            if (refl_wino_padded(d, gz))                     // [padded-grid gradient | slabs | dY in its border of zeros]
                return padded + wino_ws_floats(&gz) + align4((long)d->N * d->Cout * (d->H + 2) * (d->W + 2));
            if (refl_wino_interior(d, gz)) { const long s3 = wino_ws_floats(&gz); slabs = s3 > slabs ? s3 : slabs; }
        }
    }
    return wt + padded + slabs;
}

namespace {
int bwd_data_impl(const fd_conv_desc* d, const float* gy, const float* w, float* gx, float* wt_base, int wt_ready, float* ws,
                  void* stream, const float* gx_add = nullptr);
}
extern "C" int fd_axpby(const float* a, const float* b, float* out, long n, float alpha, float beta, void* stream);   // pool.hip
extern "C" int fd_conv2d_bwd_data(const fd_conv_desc* d, const float* gy, const float* w, float* gx, float* wt_base,
                                  int wt_ready, float* ws, void* stream) {
    return bwd_data_impl(d, gy, w, gx, wt_base, wt_ready, ws, stream);
}
extern "C" int fd_conv2d_bwd_data_add(const fd_conv_desc* d, const float* gy, const float* w, const float* gx_add, float* gx,
                                      float* wt_base, int wt_ready, float* ws, void* stream) {
    FD_REQUIRE(gx_add != gx, "fd_conv2d_bwd_data_add: gx_add must not alias gx");
    return bwd_data_impl(d, gy, w, gx, wt_base, wt_ready, ws, stream, gx_add);
}
extern "C" int fd_act_bwd(const float* y, const float* gy, float* gpre, long n, int act, void* stream);   // pool.hip
extern "C" int fd_conv2d_bwd_data_inact(const fd_conv_desc* d, const float* gy, const float* w, const float* x_in, int in_act, float* gx,
                                        float* wt_base, int wt_ready, float* ws, void* stream) {
    if (int rc = check_desc(d, "fd_conv2d_bwd_data_inact")) return rc;
    FD_REQUIRE(x_in && in_act >= 1 && in_act <= 4, "fd_conv2d_bwd_data_inact: needs the layer's input and an activation id 1..4");
    if (c1_shape_ok(d)) {                                // dispconv: the factor act'(x_in) rides in the stencil's store
        FD_REQUIRE(gy && w && gx, "fd_conv2d_bwd_data_inact: NULL tensor");
        conv_log("dgrad", "c1 stencil * act'(input)", d);
        return c1_dgrad_launch(d, gy, w, gx, (hipStream_t)stream, x_in, in_act);
    }
    if (int rc = bwd_data_impl(d, gy, w, gx, wt_base, wt_ready, ws, stream, nullptr)) return rc;
    return fd_act_bwd(x_in, gx, gx, (long)d->N * d->Cin * d->H * d->W, in_act, stream);       // every other kernel family: one element-wise pass
}
namespace {
int bwd_data_impl(const fd_conv_desc* d, const float* gy, const float* w, float* gx, float* wt_base, int wt_ready, float* ws,
                  void* stream, const float* gx_add) {
    if (int rc = check_desc(d, "fd_conv2d_bwd_data")) return rc;
    // gx_add joins in the epilogue of the MFMA kernels (and of their split-K reduction); the remaining paths (generic gather
    // GEMM, reflect padding with its fold pass, parity classes without taps) add it with one element-wise launch afterwards
    const long gx_n = (long)d->N * d->Cin * d->H * d->W;
    auto add_after = [&]() -> int { return gx_add ? fd_axpby(gx, gx_add, gx, gx_n, 1.0f, 1.0f, stream) : 0; };
    if (c1_shape_ok(d)) {                                // dispconv: a stencil with the reflect adjoint folded in (conv_c1.hip)
        FD_REQUIRE(gy && w && gx, "fd_conv2d_bwd_data: NULL tensor");
        conv_log("dgrad", "c1 stencil", d);
        if (int rc = c1_dgrad_launch(d, gy, w, gx, (hipStream_t)stream)) return rc;
        return add_after();
    }
    FD_REQUIRE(gy && w && gx && wt_base, "fd_conv2d_bwd_data: NULL tensor");
    ConvShape s;
    FD_REQUIRE(conv_out_shape(d, s), "fd_conv2d_bwd_data: empty output");
    FD_REQUIRE((long)d->N * d->Cin * (d->H + 2) * (d->W + 2) < (1L << 29) && (long)d->N * d->Cout * s.Ho * s.Wo < (1L << 29),
               "fd_conv2d_bwd_data: tensor too large for 32-bit byte offsets (2 GiB per tensor)");
    hipStream_t st = (hipStream_t)stream;
    const int KH = d->KH, KW = d->KW;
    if (limb_dgrad_ok(d)) {                              // 1x1 stride 1: gx[n][ci][p] = sum_co W[co][ci] gy[n][co][p], one GEMM with the transposed weights
        conv_log("dgrad", "limb 1x1", d);
        if (!wt_ready)
            if (int rc = limb_weight_split_launch(w, wt_base, d->Cin, d->Cout, 1, st)) return rc;
        return limb_gemm_launch(wt_base, gy, gx, nullptr, gx_add, ws, d->Cin, d->Cout, d->N, d->H * d->W, 0, st);
    }
    {
        fd_conv_desc gd;
        conv_log("dgrad", wino_dgrad_desc(d, gd) ? "wino" : (d->stride == 2 && fast_dgrad_ok(d) && limb_conv_dgrad_ok(d)) ? "limb direct" : "direct", d);
        if (wino_dgrad_desc(d, gd)) {
            if (!wt_ready)
                if (int rc = wino_weight_launch(&gd, w, wt_base, 1, st)) return rc;
            return wino_conv_launch(&gd, gy, wt_base, nullptr, gx, ws, st, gx_add);
        }
    }
    const bool fast = fast_dgrad_ok(d);
    bool add_in_kernel = false;
    const long wt_n = align4((long)d->Cin * d->Cout * KH * KW);
    float* wt = wt_base;                       // per parity class: wt_base + class * wt_n
    float* gpad = ws;
    const long pad_n = d->pad_mode == 1 ? align4((long)d->N * d->Cin * (d->H + 2) * (d->W + 2)) : 0;
    float* slabs = ws ? ws + pad_n : nullptr;
    FD_REQUIRE(ws || (pad_n == 0), "fd_conv2d_bwd_data: workspace required for reflect padding");

    // common geometry of "a conv over gy": channels = Cout, spatial = Ho x Wo
    GemmArgs g = {};
    g.X = gy; g.bias = nullptr; g.act = 0; g.in_norm = 0; g.pad_mode = 0;
    g.M = d->Cin; g.Nb = d->N; g.C = d->Cout; g.Hi = s.Ho; g.Wi = s.Wo;
    auto run = [&](int TA, int TB, int kh0, int dkh, int kw0, int dkw, bool allow_split) -> int {
        if (fast) {
            if (!wt_ready)
                if (int rc = fast_weight_relayout(w, wt, d->Cout, d->Cin, KH, KW, TA, TB, kh0, dkh, kw0, dkw, 1, st)) return rc;
            FastGemmArgs f = {};
            f.A = wt; f.X = gy; f.Y = g.Y; f.bias = nullptr;
            f.M = d->Cin; f.C = d->Cout; f.T = TA * TB; f.TB = TB; f.K = f.T * f.C;
            f.Nb = d->N; f.Hi = s.Ho; f.Wi = s.Wo; f.NY = g.NY; f.NX = g.NX;
            f.sy = g.sy; f.oy = g.oy; f.da = g.da; f.sx = g.sx; f.ox = g.ox; f.db = g.db;
            f.pad_mode = 0;
            f.out_ns = g.out_ns; f.out_cs = g.out_cs; f.out_w = g.out_w;
            f.osy = g.osy; f.ooy = g.ooy; f.osx = g.osx; f.oox = g.oox;
            f.out_total = (long)d->N * g.out_ns; f.slab_stride = f.out_total;
            f.slabs = slabs;       // split-K is only chosen for unit-stride outputs (fast_splitk_slab_floats)
            f.add = add_in_kernel ? gx_add : nullptr;
            (void)allow_split;
            return fast_gemm_launch(f, st);
        }
        if (!wt_ready) {
            hipLaunchKernelGGL(k_weight_relayout, dim3(ew_blocks((long)d->Cin * d->Cout * TA * TB)), dim3(256), 0, st, w, wt,
                               d->Cout, d->Cin, KH, KW, TA, TB, kh0, dkh, kw0, dkw);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_data(relayout)");
        }
        g.A = wt; g.K = d->Cout * TA * TB;
        if (int rc = dispatch_gemm(TA, TB, g, st)) return rc;
        FD_LAUNCH_CHECK("fd_conv2d_bwd_data(gemm)");
        return 0;
    };

    if (d->stride == 1) {
        g.sy = 1; g.da = 1; g.sx = 1; g.db = 1;
        g.osy = 1; g.ooy = 0; g.osx = 1; g.oox = 0;
        const int ring_on = fd_tun().reflect_ring;
        fd_conv_desc gz;
        if (d->pad_mode == 1 && refl_wino_padded(d, gz)) {
            conv_log("dgrad", "wino on the padded grid + fold", d);
            const long planes_in = (long)d->N * d->Cout, planes_out = (long)d->N * d->Cin;
            const long np = (long)(d->H + 2) * (d->W + 2);
            float* wslabs = ws + pad_n;
            float* gyp = wslabs + wino_ws_floats(&gz);
            if (!wt_ready)
                if (int rc = wino_weight_launch(&gz, w, wt, 1, st)) return rc;
            const long bx = (np + 255) / 256;
            hipLaunchKernelGGL(k_zero_border_copy, dim3((unsigned)(bx > 64 ? 64 : bx), (unsigned)(planes_in > 32768 ? 32768 : planes_in)), dim3(256), 0, st,
                               gy, gyp, planes_in, d->H, d->W);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_data(zero border)");
            if (int rc = wino_conv_launch(&gz, gyp, wt, nullptr, gpad, wslabs, st, nullptr)) return rc;
            const long fold_bx = ((long)d->H * d->W + 255) / 256;
            hipLaunchKernelGGL(k_reflect_fold, dim3((unsigned)(fold_bx > 64 ? 64 : fold_bx), (unsigned)(planes_out > 32768 ? 32768 : planes_out)),
                               dim3(256), 0, st, gpad, gx, planes_out, d->H, d->W);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_data(fold)");
            return add_after();
        }
        const bool wino_interior = refl_wino_interior(d, gz);
        if (d->pad_mode == 1 && fast && ring_on && KH == 3 && KW == 3 && d->pad == 1 && d->H >= 2 && d->W >= 2 &&
            (wino_interior || (long)d->H * d->W >= (ring_on > 1 ? ring_on : 16384))) {      // smaller planes (measured up to 48 x 160): four thin launches + their fold cost more than the fold pass
            // Reflect padding, 3x3: (1) the interior of the padded grid = the zero-padded data gradient, straight into gx (with the
            // second gradient of the tensor, if any, in the epilogue); (2) the ring's four strips as ONE grouped launch of thin
            // problems into a small buffer; (3) k_reflect_ring_fold.  The padded-grid gradient + k_reflect_fold of rounds 1-2 wrote
            // and re-read the whole (H+2) x (W+2) tensor on the decoder's serial chain (0.65 ms per training step).
            g.NY = d->H; g.NX = d->W; g.oy = -1; g.ox = -1;
            g.Y = gx; g.out_w = d->W; g.out_cs = (long)d->H * d->W; g.out_ns = g.out_cs * d->Cin;
            add_in_kernel = false;       // a second gradient of the tensor (not used by the decoder) is added after the fold, so
                                         // that the sum keeps the order (interior + ring) + other of the fold path, bit for bit
            if (wino_interior) {                         // the decoder's wide blocks: the interior on the Winograd kernels
                conv_log("dgrad", "wino + ring", d);
                float* wt_wino = wt + wt_n;
                if (!wt_ready) {
                    if (int rc = fast_weight_relayout(w, wt, d->Cout, d->Cin, KH, KW, KH, KW, KH - 1, -1, KW - 1, -1, 1, st)) return rc;
                    if (int rc = wino_weight_launch(&gz, w, wt_wino, 1, st)) return rc;
                }
                if (int rc = wino_conv_launch(&gz, gy, wt_wino, nullptr, gx, slabs, st, nullptr)) return rc;
            } else if (n16_shape_ok(d, d->Cin, d->Cout)) {      // the zero-padded data gradient of a 16 / 32-channel block: conv_n16.hip on dY
                conv_log("dgrad", "n16 + ring", d);
                if (!wt_ready)                           // the ring below still runs on the implicit-GEMM kernel and its layout
                    if (int rc = fast_weight_relayout(w, wt, d->Cout, d->Cin, KH, KW, KH, KW, KH - 1, -1, KW - 1, -1, 1, st)) return rc;
                if (int rc = n16_launch(d, d->Cin, d->Cout, gy, w, nullptr, gx, 1, 0, 0, st)) return rc;
            } else if (int rc = run(KH, KW, KH - 1, -1, KW - 1, -1, true)) return rc;
            const int Hp = d->H, Wp2 = d->W + 2;
            const long ring_plane = 2L * Wp2 + 2L * Hp;
            float* ring = gpad;                                   // [N][Cin][top W+2 | bottom W+2 | left H | right H]
            FastGemmArgs f = {};
            FastGemmGroup q = {};
            f.A = wt; f.X = gy; f.Y = ring; f.bias = nullptr;
            f.M = d->Cin; f.C = d->Cout; f.T = 9; f.TB = 3; f.K = 9 * d->Cout;
            f.Nb = d->N; f.Hi = s.Ho; f.Wi = s.Wo;
            f.sy = 1; f.da = 1; f.sx = 1; f.db = 1; f.pad_mode = 0;       // the flip is in the weight layout (kh0 = 2, dkh = -1)
            f.osy = 1; f.osx = 1; f.ooy = 0; f.oox = 0;
            f.out_total = (long)d->N * d->Cin * ring_plane; f.slab_stride = f.out_total; f.slabs = nullptr; f.add = nullptr;
            q.n = 4; q.own_out = 1;
            const int ny[4] = {1, 1, Hp, Hp}, nx[4] = {Wp2, Wp2, 1, 1};
            const int oy4[4] = {-2, -2 + d->H + 1, -1, -1}, ox4[4] = {-2, -2, -2, -2 + d->W + 1};
            const long yoff[4] = {0, Wp2, 2L * Wp2, 2L * Wp2 + Hp};
            for (int j = 0; j < 4; ++j) {
                q.A[j] = wt; q.NY[j] = ny[j]; q.NX[j] = nx[j]; q.oy[j] = oy4[j]; q.ox[j] = ox4[j]; q.ooy[j] = 0; q.oox[j] = 0;
                q.T[j] = 9; q.TB[j] = 3; q.K[j] = 9 * d->Cout;
                q.y_off[j] = yoff[j]; q.out_w[j] = nx[j]; q.out_cs[j] = ring_plane; q.out_ns[j] = ring_plane * d->Cin;
            }
            f.NY = q.NY[0]; f.NX = q.NX[0]; f.oy = q.oy[0]; f.ox = q.ox[0];
            f.out_w = q.out_w[0]; f.out_cs = q.out_cs[0]; f.out_ns = q.out_ns[0];
            if (int rc = fast_gemm_group_launch(f, q, st)) return rc;
            const long planes = (long)d->N * d->Cin;
            const int rows2 = (d->H - 2 != 1) ? 2 : 1, cols2 = (d->W - 2 != 1) ? 2 : 1;
            const long targets = planes * ((long)rows2 * d->W + (long)(d->H - rows2) * cols2);
            hipLaunchKernelGGL(k_reflect_ring_fold, dim3((unsigned)ew_blocks(targets)), dim3(256), 0, st, ring, gx, planes, d->H, d->W);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_data(ring fold)");
            return add_after();
        }
        if (d->pad_mode == 1) {   // gradient on the reflect-padded grid, then fold (adjoint of ReflectionPad2d(1))
            g.NY = d->H + 2; g.NX = d->W + 2; g.oy = -(KH - 1); g.ox = -(KW - 1);
            g.Y = gpad; g.out_w = d->W + 2;
            g.out_cs = (long)(d->H + 2) * (d->W + 2); g.out_ns = g.out_cs * d->Cin;
            if (int rc = run(KH, KW, KH - 1, -1, KW - 1, -1, true)) return rc;
            const long n = (long)d->N * d->Cin * d->H * d->W;
            const long fold_planes = (long)d->N * d->Cin, fold_bx = ((long)d->H * d->W + 255) / 256;
            hipLaunchKernelGGL(k_reflect_fold, dim3((unsigned)(fold_bx > 64 ? 64 : fold_bx), (unsigned)(fold_planes > 32768 ? 32768 : fold_planes)),
                               dim3(256), 0, st, gpad, gx, fold_planes, d->H, d->W);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_data(fold)");
            return add_after();
        }
        g.NY = d->H; g.NX = d->W; g.oy = -(KH - 1 - d->pad); g.ox = -(KW - 1 - d->pad);
        g.Y = gx; g.out_w = d->W; g.out_cs = (long)d->H * d->W; g.out_ns = g.out_cs * d->Cin;
        add_in_kernel = fast && gx_add;
        if (int rc = run(KH, KW, KH - 1, -1, KW - 1, -1, true)) return rc;
        return add_in_kernel ? 0 : add_after();
    }
    // stride 2: four output-parity classes, each a dense conv over its own tap subset
    g.out_w = d->W; g.out_cs = (long)d->H * d->W; g.out_ns = g.out_cs * d->Cin; g.Y = gx;
    bool need_zero = false;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw)
            if (((ph + d->pad) & 1) >= KH || ((pw + d->pad) & 1) >= KW) need_zero = true;
    if (need_zero) {
        if (hipMemsetAsync(gx, 0, sizeof(float) * (size_t)d->N * d->Cin * d->H * d->W, st) != hipSuccess) {
            fd_set_error("fd_conv2d_bwd_data: memset failed");
            return -1;
        }
    }
    add_in_kernel = fast && gx_add && !need_zero;       // every element of gx is written by exactly one parity class
    if (fast) {
        // the (up to) four classes in ONE launch: alone, a class has a quarter of the pixels and no split-K (its output is strided) -
        // 92 workgroups for layer4.0 at batch 24; four launches in a row measured 253 us there (27 TFLOP/s)
        FastGemmArgs f = {};
        FastGemmGroup q = {};
        f.X = gy; f.Y = gx; f.bias = nullptr;
        f.M = d->Cin; f.C = d->Cout;
        f.Nb = d->N; f.Hi = s.Ho; f.Wi = s.Wo;
        f.sy = 1; f.da = -1; f.sx = 1; f.db = -1;
        f.pad_mode = 0;
        f.out_ns = g.out_ns; f.out_cs = g.out_cs; f.out_w = g.out_w;
        f.osy = 2; f.osx = 2;
        f.out_total = (long)d->N * g.out_ns; f.slab_stride = f.out_total; f.slabs = nullptr;
        f.add = add_in_kernel ? gx_add : nullptr;
        // the classes' weights: [Cin][(tap, Cout)] fp32 for k_conv_fast_grp, or its pre-split image for k_conv_limb_grp (in the same slots:
        // 1.5 x (<= 4 of 9 taps) of a slot; a 1x1 kernel has one class and four slots)
        const bool limb_s2 = limb_conv_dgrad_ok(d);
        // (classes with the most taps first: their workgroups are the launch's longest - 4, 2, 2, 1 taps for a 3x3 kernel with pad 1)
        for (int ph = 1; ph >= 0; --ph)
            for (int pw = 1; pw >= 0; --pw) {
                const int kh0 = (ph + d->pad) & 1, kw0 = (pw + d->pad) & 1;
                if (kh0 >= KH || kw0 >= KW) continue;
                const int TA = (KH - kh0 + 1) / 2, TB = (KW - kw0 + 1) / 2;
                const int NY = (d->H - ph + 1) / 2, NX = (d->W - pw + 1) / 2;
                if (NY <= 0 || NX <= 0) continue;
                float* wc = wt_base + (long)(ph * 2 + pw) * wt_n;
                if (!wt_ready) {
                    if (limb_s2) { if (int rc = limb_conv_weight_split_launch(w, wc, d->Cout, d->Cin, KH, KW, TA, TB, kh0, 2, kw0, 2, 1, st)) return rc; }
                    else if (int rc = fast_weight_relayout(w, wc, d->Cout, d->Cin, KH, KW, TA, TB, kh0, 2, kw0, 2, 1, st)) return rc;
                }
                const int j = q.n++;
                q.A[j] = wc; q.NY[j] = NY; q.NX[j] = NX;
                q.oy[j] = (ph + d->pad - kh0) / 2; q.ox[j] = (pw + d->pad - kw0) / 2;
                q.ooy[j] = ph; q.oox[j] = pw;
                q.T[j] = TA * TB; q.TB[j] = TB; q.K[j] = TA * TB * d->Cout;
            }
        if (q.n > 0) {
            f.A = q.A[0]; f.NY = q.NY[0]; f.NX = q.NX[0]; f.T = q.T[0]; f.TB = q.TB[0]; f.K = q.K[0];
            if (limb_s2) { if (int rc = limb_conv_group_launch(f, q, st)) return rc; }
            else if (int rc = fast_gemm_group_launch(f, q, st)) return rc;
        }
        return add_in_kernel ? 0 : add_after();
    }
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {
            const int kh0 = (ph + d->pad) & 1, kw0 = (pw + d->pad) & 1;
            if (kh0 >= KH || kw0 >= KW) continue;
            const int TA = (KH - kh0 + 1) / 2, TB = (KW - kw0 + 1) / 2;
            const int NY = (d->H - ph + 1) / 2, NX = (d->W - pw + 1) / 2;
            if (NY <= 0 || NX <= 0) continue;
            g.NY = NY; g.NX = NX;
            g.sy = 1; g.oy = (ph + d->pad - kh0) / 2; g.da = -1;
            g.sx = 1; g.ox = (pw + d->pad - kw0) / 2; g.db = -1;
            g.osy = 2; g.ooy = ph; g.osx = 2; g.oox = pw;
            wt = wt_base + (long)(ph * 2 + pw) * wt_n;
            if (int rc = run(TA, TB, kh0, 2, kw0, 2, false)) return rc;
        }
    return add_in_kernel ? 0 : add_after();
}
}  // namespace

namespace {
// re-layout mode of a Winograd layout: `g` = the convolution the kernel computes (for a data gradient: channels already swapped)
inline int wino_layout_mode(const fd_conv_desc* g, bool dgrad) {
    if (wino_fwd_limb(g)) return dgrad ? 12 : 11;
    if (wino_fwd_2d(g)) return dgrad ? 6 : 5;
    return dgrad ? 4 : 3;
}
}  // namespace
extern "C" int fd_conv2d_relayout_jobs(const fd_conv_desc* d, int kind, const float* w, float* wt, fd_relayout_job* jobs) {
    if (check_desc(d, "fd_conv2d_relayout_jobs") || !w || !wt || !jobs) return 0;
    auto fill = [&](fd_relayout_job& j, float* dst, int TA, int TB, int kh0, int dkh, int kw0, int dkw, int mode) {
        j = fd_relayout_job{};
        j.w = w; j.dst = dst; j.Co = d->Cout; j.Ci = d->Cin; j.KH = d->KH; j.KW = d->KW;
        j.TA = TA; j.TB = TB; j.kh0 = kh0; j.dkh = dkh; j.kw0 = kw0; j.dkw = dkw; j.mode = mode;
        j.n = (long)d->Cout * d->Cin * TA * TB;
    };
    if (c1_shape_ok(d)) return 0;                        // stencil kernels: no layouts
    if (kind == 0) {
        if (!fast_fwd_ok(d) || n16_shape_ok(d, d->Cout, d->Cin)) return 0;
        if (limb_fwd_ok(d)) { fill(jobs[0], wt, 1, 1, 0, 1, 0, 1, 7); return 1; }
        fill(jobs[0], wt, d->KH, d->KW, 0, 1, 0, 1, wino_use_fwd(d) ? wino_layout_mode(d, false) : (limb_conv_fwd_ok(d) ? 9 : 0));
        return 1;
    }
    const int KH = d->KH, KW = d->KW;
    if (limb_dgrad_ok(d)) { fill(jobs[0], wt, 1, 1, 0, 1, 0, 1, 8); return 1; }
    {
        fd_conv_desc gd;
        if (wino_dgrad_desc(d, gd)) { fill(jobs[0], wt, KH, KW, 0, 1, 0, 1, wino_layout_mode(&gd, true)); return 1; }
    }
    const int mode = fast_dgrad_ok(d) ? 1 : 2;
    if (d->stride == 1) {
        fd_conv_desc gz;
        if (refl_wino_padded(d, gz)) { fill(jobs[0], wt, KH, KW, 0, 1, 0, 1, wino_layout_mode(&gz, true)); return 1; }
        fill(jobs[0], wt, KH, KW, KH - 1, -1, KW - 1, -1, mode);
        if (refl_wino_interior(d, gz)) {                 // second layout behind the first: U of the interior's Winograd kernel
            fill(jobs[1], wt + align4((long)d->Cin * d->Cout * KH * KW), KH, KW, 0, 1, 0, 1, wino_layout_mode(&gz, true));
            return 2;
        }
        return 1;
    }
    const long wt_n = align4((long)d->Cin * d->Cout * KH * KW);
    int n = 0;
    for (int ph = 0; ph < 2; ++ph)
        for (int pw = 0; pw < 2; ++pw) {                 // same enumeration as fd_conv2d_bwd_data
            const int kh0 = (ph + d->pad) & 1, kw0 = (pw + d->pad) & 1;
            if (kh0 >= KH || kw0 >= KW) continue;
            const int TA = (KH - kh0 + 1) / 2, TB = (KW - kw0 + 1) / 2;
            if ((d->H - ph + 1) / 2 <= 0 || (d->W - pw + 1) / 2 <= 0) continue;
            fill(jobs[n++], wt + (long)(ph * 2 + pw) * wt_n, TA, TB, kh0, 2, kw0, 2, (mode == 1 && limb_conv_dgrad_ok(d)) ? 10 : mode);
        }
    return n;
}

extern "C" long fd_relayout_plan(fd_relayout_job* jobs, int n) {
    long blocks = 0;
    for (int i = 0; i < n; ++i) { jobs[i].first_block = blocks; blocks += relayout_units(jobs[i]); }
    return blocks;
}

extern "C" int fd_relayout_batch(const fd_relayout_job* jobs_dev, int n, long total_blocks, void* stream) {
    FD_REQUIRE(jobs_dev && n > 0 && total_blocks > 0 && total_blocks < (1L << 31), "fd_relayout_batch: bad args");
    hipLaunchKernelGGL(k_relayout_batch, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_dev, n);
    FD_LAUNCH_CHECK("fd_relayout_batch");
    return 0;
}

namespace {
int wgrad_splits(const fd_conv_desc* d, const ConvShape& s) {
    const long Np = (long)d->N * s.Ho * s.Wo;
    const long J = (long)d->Cin * d->KH * d->KW;
    const long tiles = J <= 64 ? (long)fd_cdiv(d->Cout, 64) : (long)fd_cdiv(J, 128) * fd_cdiv(d->Cout, d->Cout <= 32 ? 32 : 64);
    long want = (768 + tiles - 1) / tiles;            // ~3 workgroups per CU
    long maxs = (Np + 511) / 512;                     // at least 512 pixels per split
    long sp = want < maxs ? want : maxs;
    if (sp < 1) sp = 1;
    if (sp > 96) sp = 96;
    return (int)sp;
}
}  // namespace

extern "C" long fd_conv2d_bwd_weight_ws_floats(const fd_conv_desc* d) {
    if (!d) return 0;
    ConvShape s;
    if (!conv_out_shape(d, s)) return 0;
    const long wsz = (long)d->Cout * d->Cin * d->KH * d->KW;
    long slabs;
    if (narrow_wgrad_ok(d)) slabs = narrow_wgrad_ws_floats(d);
    else if (stem_wgrad_ok(d)) slabs = stem_wgrad_ws_floats(d);
    else if (wino_use_wgrad(d)) slabs = wino_wgrad_ws_floats(d);
    else if (fast_wgrad_ok(d)) slabs = (long)fast_wgrad_splits(d->Cout, d->Cin, d->KH * d->KW, (long)d->N * s.Ho * s.Wo) * wsz;
    else { const int sp = wgrad_splits(d, s); slabs = sp > 1 ? (long)sp * wsz : 0; }
    if (limb_wgrad_ok(d)) { const long l = limb_wgrad_ws_floats(d->Cout, d->Cin, d->N, d->H * d->W); slabs = l > slabs ? l : slabs; }   // (the direct kernel stays the fallback for unaligned tensors)
    if (limb_conv_wgrad_ok(d, s)) { const long l = limb_wgrad_s2_ws_floats(d->Cout, d->Cin, d->N, s.Ho * s.Wo, d->KH * d->KW); slabs = l > slabs ? l : slabs; }
    const long bias_part = (long)d->Cout * CS_SPLITS;
    if (slabs < wsz) slabs = wsz;                            // accumulate mode stages a single slab
    return slabs > bias_part ? slabs : bias_part;          // the two uses are sequential on the stream
}

extern "C" int fd_conv2d_bwd_weight(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* gbias,
                                    float* ws, int accumulate, void* stream) {
    if (int rc = check_desc(d, "fd_conv2d_bwd_weight")) return rc;
    FD_REQUIRE(x && gy && gw && ws, "fd_conv2d_bwd_weight: NULL tensor / workspace");
    ConvShape s;
    FD_REQUIRE(conv_out_shape(d, s), "fd_conv2d_bwd_weight: empty output");
    FD_REQUIRE((long)d->N * d->Cin * d->H * d->W < (1L << 29) && (long)d->N * d->Cout * s.Ho * s.Wo < (1L << 29),
               "fd_conv2d_bwd_weight: tensor too large for 32-bit byte offsets (2 GiB per tensor)");
    hipStream_t st = (hipStream_t)stream;
    const long Np = (long)d->N * s.Ho * s.Wo;
    const bool limb_w = limb_wgrad_ok(d) && (((uintptr_t)x | (uintptr_t)gy) & 15) == 0;
    const bool limb_w2 = !limb_w && limb_conv_wgrad_ok(d, s) && (((uintptr_t)x | (uintptr_t)gy) & 15) == 0;
    conv_log("wgrad", limb_w ? "limb 1x1" : limb_w2 ? "limb direct" : narrow_wgrad_ok(d) ? "narrow" : stem_wgrad_ok(d) ? "stem" : wino_use_wgrad(d) ? "wino" : fast_wgrad_ok(d) ? "direct" : "generic", d);
    if (limb_w) {
        if (int rc = limb_wgrad_launch(x, gy, gw, ws, d->Cout, d->Cin, d->N, d->H * d->W, accumulate, st)) return rc;
    } else if (limb_w2) {
        if (int rc = limb_wgrad_s2_launch(x, gy, gw, ws, d->Cout, d->Cin, d->N, d->H, d->W, s.Ho, s.Wo, d->KH * d->KW, accumulate, st)) return rc;
    } else if (narrow_wgrad_ok(d)) {
        if (int rc = narrow_wgrad_launch(d, x, gy, gw, ws, accumulate, st)) return rc;
    } else if (stem_wgrad_ok(d)) {
        if (int rc = stem_wgrad_launch(d, x, gy, gw, ws, accumulate, st)) return rc;
    } else if (wino_use_wgrad(d)) {
        if (int rc = wino_wgrad_launch(d, x, gy, gw, ws, accumulate, st)) return rc;
    } else if (fast_wgrad_ok(d)) {
        FastWgradArgs f = {};
        f.dY = gy; f.X = x; f.slabs = ws;
        f.M = d->Cout; f.C = d->Cin; f.T = d->KH * d->KW; f.TB = d->KW;
        f.Nb = d->N; f.Hi = d->H; f.Wi = d->W; f.NY = s.Ho; f.NX = s.Wo;
        f.sy = d->stride; f.oy = -d->pad; f.da = 1; f.sx = d->stride; f.ox = -d->pad; f.db = 1;
        f.pad_mode = d->pad_mode;
        f.dy_cs = (long)s.Ho * s.Wo; f.dy_ns = f.dy_cs * d->Cout;
        if (int rc = fast_wgrad_launch(f, gw, fast_wgrad_splits(f.M, f.C, f.T, Np), accumulate, st)) return rc;
    } else {
        const int sp = wgrad_splits(d, s);
        const bool staged = sp > 1 || accumulate;
        WgradArgs g = {};
        g.dY = gy; g.X = x; g.out = staged ? ws : gw;
        g.M = d->Cout; g.J = d->Cin * d->KH * d->KW;
        g.Nb = d->N; g.C = d->Cin; g.Hi = d->H; g.Wi = d->W;
        g.NY = s.Ho; g.NX = s.Wo;
        g.sy = d->stride; g.oy = -d->pad; g.da = 1; g.sx = d->stride; g.ox = -d->pad; g.db = 1;
        g.pad_mode = d->pad_mode; g.in_norm = d->in_norm;
        g.dy_cs = (long)s.Ho * s.Wo; g.dy_ns = g.dy_cs * d->Cout;
        long pps = (Np + sp - 1) / sp;
        pps = (pps + 31) / 32 * 32;
        g.pix_per_split = pps;
        if (int rc = dispatch_wgrad(d->KH, d->KW, g, sp, st)) return rc;
        FD_LAUNCH_CHECK("fd_conv2d_bwd_weight");
        if (staged) {
            const long n = (long)g.M * g.J;
            hipLaunchKernelGGL(k_reduce_slabs, dim3(ew_blocks(n)), dim3(256), 0, st, ws, gw, n, sp, accumulate);
            FD_LAUNCH_CHECK("fd_conv2d_bwd_weight(reduce)");
        }
    }
    if (gbias) {
        const long plane = (long)s.Ho * s.Wo;
        const bool vec = (plane & 3) == 0 && ((uintptr_t)gy & 15) == 0;
        hipLaunchKernelGGL((vec ? k_channel_sum_part<true> : k_channel_sum_part<false>), dim3(d->Cout, CS_SPLITS), dim3(256), 0, st, gy, ws, d->N,
                           d->Cout, plane);
        FD_LAUNCH_CHECK("fd_conv2d_bwd_weight(bias)");
        hipLaunchKernelGGL(k_channel_sum_fin, dim3(fd_cdiv(d->Cout, 64)), dim3(64), 0, st, ws, gbias, d->Cout, accumulate);
        FD_LAUNCH_CHECK("fd_conv2d_bwd_weight(bias fin)");
    }
    return 0;
}
