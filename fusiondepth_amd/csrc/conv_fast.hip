// Fast path of the FP32 MFMA implicit-GEMM convolution for channel counts that are multiples of 16 (every
// ResNet / decoder / pose conv except the 7x7 stems, 1-channel dispconv gradients and 12-channel pose output).
//
// What changed w.r.t. the generic kernel in conv.hip (measured there: ~19 VALU per MFMA, 25-40 TFLOP/s):
//   * GEMM-K is ordered (tap, channel) instead of (channel, tap): a K-chunk = ONE tap x BKC consecutive channels,
//     so the tap's bounds / reflect logic runs once per chunk and the BKC loads of a thread are
//     base + i * stride (one v_add each) with a wave-uniform base pointer (SGPR) + 32-bit lane offsets;
//   * each wave owns up to 2x2 accumulator tiles of 32x32 (64x64 outputs) => 4 MFMAs per 4 LDS operand reads,
//     64 MFMAs (4096 SIMD cycles) per barrier at BKC = 32, enough to cover the HBM latency of the next chunk;
//   * weights are re-laid-out to [M][tap][channel] by a tiny prep kernel and loaded as float4;
//   * small-N layers (layer3/4 at micro-batch 6: N = 2880 / 720 pixels) are split over K into deterministic
//     slabs so that >= 256 workgroups exist; a finishing kernel sums the slabs and applies bias + activation.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>

#ifndef FD_LS_DIV
#define FD_LS_DIV 2
#endif
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ELU / sigmoid / tanh: ONE out-of-line copy (the unrolled epilogues would otherwise inline three libm routines per accumulator)
__device__ __attribute__((noinline)) float act_apply_slow(float v, int act) {
    if (act == 2) return v > 0.f ? v : expm1f(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    return tanhf(v);
}
__device__ __forceinline__ float act_apply(float v, int act) {
    if (act >= 2) return act_apply_slow(v, act);
    return act == 1 ? (v > 0.f ? v : 0.f) : v;
}
__device__ __forceinline__ int refl_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// The kernel body; `bx` = pixel tile of problem `g`, `ntx` = its number of pixel tiles (k_conv_fast: the grid's x extent;
// k_conv_fast_grp: the x range of one of several problems that share a launch).
template <int WAVES_M, int WAVES_N, int WM, int WN, int BKC>
__device__ __forceinline__ void conv_fast_body(const FastGemmArgs& g, int bx, int ntx) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    constexpr int LDA = BM + 1, LDB = BN;
    constexpr int RP = NT / BN;                 // channel rows per pass of the activation loader
    constexpr int NB_LOAD = BKC / RP;
    constexpr int A_V4_PER_ROW = BKC / 4;       // float4 per weight row per chunk
    constexpr int A_ROWS_PER_PASS = NT / A_V4_PER_ROW;
    constexpr int NA_LOAD = (BM + A_ROWS_PER_PASS - 1) / A_ROWS_PER_PASS;
    constexpr bool A_PARTIAL = (BM % A_ROWS_PER_PASS) != 0;      // fewer weight rows than loader rows: some threads idle
    static_assert(NT % BN == 0 && BKC % RP == 0, "tile/loader mismatch");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                           // [2][BKC * LDA]
    float* sB = smem + 2 * BKC * LDA;           // [2][BKC * LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    if (g.xcd_swizzle) { const int per = ntx >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const int m0 = blockIdx.y * BM;
    const long p0 = (long)bx * BN;
    const int plane = g.NY * g.NX;
    const long Np = (long)g.Nb * plane;
    const unsigned chw = (unsigned)(g.Hi * g.Wi);
    const int cpt = g.C / BKC;                  // chunks per tap
    const int nchunk_all = g.T * cpt;
    // split-K: this workgroup handles chunks [ch_lo, ch_hi)
    const int nsplit = (int)gridDim.z, zs = (int)blockIdx.z;
    const int per_split = (nchunk_all + nsplit - 1) / nsplit;
    const int ch_lo = zs * per_split;
    const int ch_hi = ch_lo + per_split < nchunk_all ? ch_lo + per_split : nchunk_all;

    // ---- activation loader: fixed pixel column, channel rows kr + RP*i
    const int jn = tid % BN;
    const int kr = __builtin_amdgcn_readfirstlane(tid / BN);
    const long pg = p0 + jn;
    const bool pvalid = pg < Np;
    int ry0, cx0;
    unsigned nbase;
    {
        const long pp = pvalid ? pg : 0;
        const int n = (int)(pp / plane);
        const int rem = (int)(pp - (long)n * plane);
        const int y = rem / g.NX, x = rem - y * g.NX;
        ry0 = y * g.sy + g.oy; cx0 = x * g.sx + g.ox;
        nbase = (unsigned)n * (unsigned)g.C * chw;
    }
    // ---- weight loader: float4 column a4 of row ar + A_ROWS_PER_PASS*i
    const int a4 = tid % A_V4_PER_ROW, ar = tid / A_V4_PER_ROW;

    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.A), rsX = fd_make_rsrc(g.X);
    float4 ra[NA_LOAD];
    float rb[NB_LOAD];
    // state of the chunk being fetched (set by prep_chunk, consumed by the load/store slices); byte offsets
    unsigned a_off[NA_LOAD], b_off = FD_OOB;
    const unsigned b_step = 4u * (unsigned)RP * chw;                  // wave-uniform
    // (tap row, tap column, first channel) of the next chunk to prepare: advanced incrementally, all wave-uniform
    int pc_ta, pc_tb, pc_c0;
    { const int t = ch_lo / cpt; pc_c0 = (ch_lo - t * cpt) * BKC; pc_ta = t / g.TB; pc_tb = t - pc_ta * g.TB; }
    const bool refl = g.pad_mode == 1;
    auto prep_chunk = [&](bool live) __attribute__((always_inline)) {
        const unsigned k0 = (unsigned)(pc_ta * g.TB + pc_tb) * (unsigned)g.C + (unsigned)pc_c0;
#pragma unroll
        for (int i = 0; i < NA_LOAD; ++i) {
            int m = m0 + ar + A_ROWS_PER_PASS * i;
            m = m < g.M ? m : g.M - 1;                                // rows >= M are never stored by the epilogue
            a_off[i] = live ? 4u * ((unsigned)m * (unsigned)g.K + k0 + 4u * a4) : FD_OOB;
        }
        int r = ry0 + pc_ta * g.da, cc = cx0 + pc_tb * g.db;
        const bool inb = ((unsigned)r < (unsigned)g.Hi) & ((unsigned)cc < (unsigned)g.Wi);
        const int rr = refl_idx(r, g.Hi), rc = refl_idx(cc, g.Wi);
        r = refl ? rr : r; cc = refl ? rc : cc;
        const bool ok = pvalid & live & (refl | inb);
        b_off = ok ? 4u * (nbase + (unsigned)(pc_c0 + kr) * chw + (unsigned)(r * g.Wi + cc)) : FD_OOB;
        pc_c0 += BKC;
        if (pc_c0 >= g.C) { pc_c0 = 0; ++pc_tb; if (pc_tb >= g.TB) { pc_tb = 0; ++pc_ta; } }
    };
    auto load_a = [&](int i) __attribute__((always_inline)) {
        if (!A_PARTIAL || ar + A_ROWS_PER_PASS * i < BM) ra[i] = fd_ldg128(rsA, a_off[i]);
    };
    auto load_b = [&](int i) __attribute__((always_inline)) { rb[i] = fd_ldg32(rsX, b_off + (unsigned)i * b_step); };
    auto store_a = [&](int buf, int i) __attribute__((always_inline)) {
        if (A_PARTIAL && ar + A_ROWS_PER_PASS * i >= BM) return;
        float* q = sA + buf * BKC * LDA + (4 * a4) * LDA + ar + A_ROWS_PER_PASS * i;
        q[0] = ra[i].x; q[LDA] = ra[i].y; q[2 * LDA] = ra[i].z; q[3 * LDA] = ra[i].w;
    };
    auto store_b = [&](int buf, int i) __attribute__((always_inline)) {
        sB[buf * BKC * LDB + (kr + RP * i) * LDB + jn] = rb[i];
    };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NK = BKC / 2;          // MFMA k-steps per chunk
    constexpr int LS = NK / FD_LS_DIV;   // the first LS k-steps issue the next chunk's loads, the last LS store them
    const int arow = lane >> 5, acol = lane & 31;
    if (ch_lo < ch_hi) {
        prep_chunk(true);
#pragma unroll
        for (int i = 0; i < NA_LOAD; ++i) load_a(i);
#pragma unroll
        for (int i = 0; i < NB_LOAD; ++i) load_b(i);
#pragma unroll
        for (int i = 0; i < NA_LOAD; ++i) store_a(0, i);
#pragma unroll
        for (int i = 0; i < NB_LOAD; ++i) store_b(0, i);
        __syncthreads();
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            const int cur = (ch - ch_lo) & 1;
            prep_chunk(ch + 1 < ch_hi);                // past the end: every load is out of range (zeros, no traffic)
            const float* pa = sA + cur * BKC * LDA + arow * LDA + wave_m * 32 * WM + acol;
            const float* pb = sB + cur * BKC * LDB + arow * LDB + wave_n * 32 * WN + acol;
            // Fine-grained software pipeline: LDS operand reads of k-step kk+1, the global loads of chunk ch+1 (first half
            // of the k-steps) and their LDS stores into the other buffer (second half) are issued in the shadow of the
            // MFMAs of k-step kk, so the matrix pipe sees no separate load / store phases.
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
                __builtin_amdgcn_sched_barrier(0);      // keep the memory ops ahead of this step's MFMAs
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

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)).
    //      Output, residual and bias go through buffer resources with 32-bit byte offsets (every tensor of this path is < 2 GiB: the
    //      launchers check): a row >= M or a pixel past the end gets an out-of-range offset instead of a branch, the bias values of
    //      the lane's rows are fetched in one batch in front of the row loop, and the activation switch sits outside it (the first
    //      version - a bias load + wait, an inlined libm switch and 64-bit address arithmetic per accumulator - executed ~1 150 vector
    //      instructions per wave and tile, more than the chunk loop of a 3x3 layer).
    const bool final_pass = nsplit == 1;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(final_pass ? g.Y : g.slabs + (size_t)blockIdx.z * g.slab_stride);
    const __amdgpu_buffer_rsrc_t rsAdd = fd_make_rsrc(g.add ? g.add : g.Y);
    const bool has_add = final_pass && g.add;
    const unsigned cs4 = 4u * (unsigned)g.out_cs;
    float bias_r[WM][16];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) bias_r[i][r] = 0.f;
    if (final_pass && g.bias) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                bias_r[i][r] = g.bias[m < g.M ? m : g.M - 1];
            }
    }
    auto rows = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;                  // 0: none (compile time), -1: g.act at run time
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const long p = p0 + wave_n * 32 * WN + j * 32 + acol;
            unsigned pix = FD_OOB;
            if (p < Np) {
                const int n = (int)(p / plane);
                const int rem = (int)(p - (long)n * plane);
                const int y = rem / g.NX, x = rem - y * g.NX;
                pix = 4u * (unsigned)((long)n * g.out_ns + (long)(y * g.osy + g.ooy) * g.out_w + (x * g.osx + g.oox));
            }
            unsigned off[WM][16];
            float addv[WM][16];
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                    off[i][r] = (m < g.M) ? pix + (unsigned)m * cs4 : FD_OOB;              // FD_OOB + (< 2^31) stays out of range
                    addv[i][r] = 0.f;
                }
            if (has_add) {                                   // the second gradient of this tensor: all loads in flight, one wait
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) addv[i][r] = fd_ldg32(rsAdd, off[i][r]);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r];
                    if (final_pass) {
                        v += bias_r[i][r];
                        if (ACT != 0) v = act_apply(v, g.act);
                        v += addv[i][r];
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, (int)off[i][r], 0, 0);
                }
        }
    };
    if (g.act == 0 || !final_pass) rows(std::integral_constant<int, 0>{});
    else rows(std::integral_constant<int, -1>{});
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int BKC>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_conv_fast(FastGemmArgs g) {
    conv_fast_body<WAVES_M, WAVES_N, WM, WN, BKC>(g, (int)blockIdx.x, (int)gridDim.x);
}

// Several problems that differ only in the fields of FastGemmGroup - the output-parity classes of a stride-2 data gradient: each
// class alone has too few tiles to fill 256 CUs (92 workgroups for ResNet layer4.0 at batch 24) and they do not depend on each
// other, so ONE launch runs them side by side: grid x = the concatenated pixel tiles of the classes.  No split-K (gridDim.z = 1).
template <int WAVES_M, int WAVES_N, int WM, int WN, int BKC>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_conv_fast_grp(FastGemmArgs g, FastGemmGroup q) {
    int j = 0;
    while (j + 1 < q.n && (int)blockIdx.x >= q.first_bx[j + 1]) ++j;
    g.A = q.A[j]; g.NY = q.NY[j]; g.NX = q.NX[j]; g.oy = q.oy[j]; g.ox = q.ox[j]; g.ooy = q.ooy[j]; g.oox = q.oox[j];
    g.T = q.T[j]; g.TB = q.TB[j]; g.K = q.K[j];
    if (q.own_out) { g.Y += q.y_off[j]; g.out_ns = q.out_ns[j]; g.out_cs = q.out_cs[j]; g.out_w = q.out_w[j]; }
    g.xcd_swizzle = 0;
    conv_fast_body<WAVES_M, WAVES_N, WM, WN, BKC>(g, (int)blockIdx.x - q.first_bx[j], q.first_bx[j + 1] - q.first_bx[j]);
}

// Y[i] = act(sum_z slabs[z][i] + bias[channel(i)])   (fixed z order => deterministic)
__global__ void __launch_bounds__(256) k_splitk_finish(const float* __restrict__ slabs, float* __restrict__ Y,
                                                       const float* __restrict__ bias, long total, long slab_stride,
                                                       int splits, long out_cs, int M, int act, const float* __restrict__ add) {
    const unsigned cs = (unsigned)out_cs, uM = (unsigned)M;              // total < 2^31 (fast-path size guard)
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < (unsigned)total; i += gridDim.x * 256u) {
        float s = 0.f;
        int z = 0;
        for (; z + 4 <= splits; z += 4) {                        // fixed order, four loads in flight
            const float a0 = slabs[(size_t)z * slab_stride + i], a1 = slabs[(size_t)(z + 1) * slab_stride + i];
            const float a2 = slabs[(size_t)(z + 2) * slab_stride + i], a3 = slabs[(size_t)(z + 3) * slab_stride + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; z < splits; ++z) s += slabs[(size_t)z * slab_stride + i];
        if (bias) s += bias[(i / cs) % uM];
        s = act_apply(s, act);
        if (add) s += add[i];
        Y[i] = s;
    }
}

// ------------------------------------------------------------------------------------------------ wgrad
// dW2[m][t][c] = sum_p dY[m][p] * X[n(p)][c][tap t of p]; one workgroup = (tap t, BNC channels) x BM rows x
// a slice of the pixels.  GEMM-K = pixels, 32 per chunk, lanes along pixels (coalesced dY and X rows).
template <int WAVES_M, int WAVES_N, int WM, int WN>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_wgrad_fast(FastWgradArgs g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    constexpr int BP = 32;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    constexpr int RPW = NT / BP;
    constexpr int NA_LOAD = BM / RPW, NB_LOAD = BN / RPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sA = smem;                           // [2][BP * LDA]
    float* sB = smem + 2 * BP * LDA;            // [2][BP * LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int cblocks = (g.C + BN - 1) / BN;
    const int t = blockIdx.x / cblocks, c0 = (blockIdx.x - t * cblocks) * BN;
    const int ta = t / g.TB, tb = t - ta * g.TB;
    const int m0 = blockIdx.y * BM;
    const int plane = g.NY * g.NX;
    const long Np = (long)g.Nb * plane;
    const unsigned chw = (unsigned)(g.Hi * g.Wi);
    const long pbeg = (long)blockIdx.z * g.pix_per_split;
    long pend = pbeg + g.pix_per_split;
    if (pend > Np) pend = Np;

    const int pl = tid % BP, rw = tid / BP;
    const int ncol = g.C - c0 < BN ? g.C - c0 : BN;          // valid channel columns of this tile
    const int nrow = g.M - m0 < BM ? g.M - m0 : BM;

    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.dY), rsX = fd_make_rsrc(g.X);
    float ra[NA_LOAD], rb[NB_LOAD];
    // Byte offsets of this thread's dY rows / X channels.  Rows past the tile's valid range are clamped to the last valid
    // one: their products land in accumulator rows / columns the epilogue never stores.
    unsigned rowa[NA_LOAD], rowb[NB_LOAD];
#pragma unroll
    for (int i = 0; i < NA_LOAD; ++i) {
        const int r = rw + RPW * i < nrow ? rw + RPW * i : nrow - 1;
        rowa[i] = 4u * (unsigned)(m0 + r) * (unsigned)g.dy_cs;
    }
#pragma unroll
    for (int i = 0; i < NB_LOAD; ++i) {
        const int c = rw + RPW * i < ncol ? rw + RPW * i : ncol - 1;
        rowb[i] = 4u * (unsigned)(c0 + c) * chw;
    }
    unsigned offa = FD_OOB, offb = FD_OOB;
    // this lane's pixel of the next chunk: (image n, offset rem inside the plane), advanced by BP per chunk
    int pn, prem;
    { const long p = pbeg + pl; pn = (int)(p / plane); prem = (int)(p - (long)pn * plane); }
    long pcur = pbeg + pl;
    const float inv_nx = 1.0f / (float)g.NX;
    const bool refl = g.pad_mode == 1;
    auto prep_chunk = [&]() __attribute__((always_inline)) {      // pixels >= pend: everything out of range (zeros)
        const bool pv = pcur < pend;
        int y = (int)(((float)prem + 0.5f) * inv_nx);             // estimate within +-1 for planes < 2^23; fixed up below
        int x = prem - y * g.NX;
        if (x < 0) { --y; x += g.NX; }
        if (x >= g.NX) { ++y; x -= g.NX; }
        offa = pv ? 4u * ((unsigned)pn * (unsigned)g.dy_ns + (unsigned)prem) : FD_OOB;
        int r = y * g.sy + g.oy + ta * g.da, cc = x * g.sx + g.ox + tb * g.db;
        const bool inb = ((unsigned)r < (unsigned)g.Hi) & ((unsigned)cc < (unsigned)g.Wi);
        const int rr = refl_idx(r, g.Hi), rc = refl_idx(cc, g.Wi);
        r = refl ? rr : r; cc = refl ? rc : cc;
        const bool ok = pv & (refl | inb);
        offb = ok ? 4u * ((unsigned)pn * (unsigned)g.C * chw + (unsigned)(r * g.Wi + cc)) : FD_OOB;
        pcur += BP; prem += BP;
        while (prem >= plane) { prem -= plane; ++pn; }
    };
    auto load_a = [&](int i) __attribute__((always_inline)) { ra[i] = fd_ldg32(rsY, offa + rowa[i]); };
    auto load_b = [&](int i) __attribute__((always_inline)) { rb[i] = fd_ldg32(rsX, offb + rowb[i]); };
    auto store_a = [&](int buf, int i) __attribute__((always_inline)) { sA[buf * BP * LDA + pl * LDA + rw + RPW * i] = ra[i]; };
    auto store_b = [&](int buf, int i) __attribute__((always_inline)) { sB[buf * BP * LDB + pl * LDB + rw + RPW * i] = rb[i]; };

    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunk = pend > pbeg ? (int)((pend - pbeg + BP - 1) / BP) : 0;
    constexpr int NK = BP / 2, LS = NK / FD_LS_DIV;
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
            const float* pa = sA + cur * BP * LDA + arow * LDA + wave_m * 32 * WM + acol;
            const float* pb = sB + cur * BP * LDB + arow * LDB + wave_n * 32 * WN + acol;
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
    }
    // slab layout [z][m][t][c]
    float* out = g.slabs + (size_t)blockIdx.z * g.M * g.T * g.C;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int c = c0 + wave_n * 32 * WN + j * 32 + acol;
        if (c >= g.C) continue;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wave_m * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                if (m < g.M) out[((size_t)m * g.T + t) * g.C + c] = acc[i][j][r];
            }
    }
}

// gw[m][c][t] (OIHW) = sum_z slabs[z][m][t][c].  Threads enumerate the SLAB order (c fastest): the splits x n slab reads
// are coalesced, the single n-element result is written with a T-float stride.
__global__ void __launch_bounds__(256) k_wgrad_finish(const float* __restrict__ slabs, float* __restrict__ gw, int M, int C,
                                                      int T, int splits, int accumulate) {
    const unsigned n = (unsigned)M * (unsigned)C * (unsigned)T;          // < 2^31 (checked by the launcher)
    const unsigned uC = (unsigned)C, uT = (unsigned)T;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const unsigned q = i / uC, c = i - q * uC;                        // 32-bit index math: two divisions per element
        const unsigned m = q / uT, t = q - m * uT;
        const unsigned dst = (m * uC + c) * uT + t;
        float s = accumulate ? gw[dst] : 0.f;
        // fixed summation order (z ascending); four loads in flight per thread: the pass is latency-bound otherwise
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const float a0 = slabs[(size_t)z * n + i], a1 = slabs[(size_t)(z + 1) * n + i];
            const float a2 = slabs[(size_t)(z + 2) * n + i], a3 = slabs[(size_t)(z + 3) * n + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; z < splits; ++z) s += slabs[(size_t)z * n + i];
        gw[dst] = s;
    }
}

// The same reduction for 3x3 kernels with both sides coalesced: a workgroup owns (MB output channels m, 32 input channels); thread
// (r, c) sums slab row [m][r][c0 + c] over the slices (128-byte rows, fixed order, eight loads in flight - with 128 slices of a
// 64-channel layer or 8 192 (m, c block) pairs of a 512-channel one the pass was latency-bound at 4 loads and one m per workgroup),
// the R x 32 block goes through LDS, and the 288 results leave as one contiguous run of gw[m][c0 .. c0 + 31][0 .. 8] (the generic
// kernel writes single floats 36 bytes apart).  R = 9: slab rows are [ky][kx].  R = 12: slab rows are [ri][kx], the row components
// of k_wgrad_wino<.., true>, and the vertical output transform is applied here:
//   dW[ky = 0] = T0 + (T1 + T2) / 2,   dW[1] = (T1 - T2) / 2,   dW[2] = (T1 + T2) / 2 - T3.
template <int R>
__global__ void __launch_bounds__(384) k_wgrad_finish9(const float* __restrict__ slabs, float* __restrict__ gw, int M, int C,
                                                       int splits, int accumulate, int mb) {
    __shared__ float tile[12][33];
    const int c0 = blockIdx.x * 32;
    const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
    const size_t n = (size_t)M * R * C;
    const int m_hi = min(M, ((int)blockIdx.y + 1) * mb);
    const int cw = min(32, C - c0);             // the last block of a channel count that is not a multiple of 32 is narrower
    for (int m = blockIdx.y * mb; m < m_hi; ++m) {
        if (r < R && c < cw) {
            const float* p = slabs + ((size_t)m * R + r) * C + c0 + c;
            float s = 0.f;
            int z = 0;
            for (; z + 8 <= splits; z += 8) {
                float a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) a[u] = p[(size_t)(z + u) * n];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += a[u];
            }
            for (; z < splits; ++z) s += p[(size_t)z * n];
            tile[r][c] = s;
        }
        __syncthreads();
        const int j = threadIdx.x;
        if (j < 9 * cw) {
            const int cc = j / 9, tt = j - cc * 9;
            float v;
            if constexpr (R == 9) v = tile[tt][cc];
            else {
                const int ky = tt / 3, kx = tt - ky * 3;
                const float t0 = tile[kx][cc], t1 = tile[3 + kx][cc], t2 = tile[6 + kx][cc], t3 = tile[9 + kx][cc];
                const float h = 0.5f * (t1 + t2);
                v = ky == 0 ? t0 + h : (ky == 1 ? 0.5f * (t1 - t2) : h - t3);
            }
            float* o = gw + ((size_t)m * C + c0) * 9 + j;
            *o = (accumulate ? *o : 0.f) + v;
        }
        __syncthreads();
    }
}

// A2[m][(a,b)][c]:  mode 0 (forward)  A2[co][t][ci] = W[co][ci][kh0+dkh*a][kw0+dkw*b]
//                   mode 1 (dgrad)    A2[ci][t][co] = W[co][ci][kh0+dkh*a][kw0+dkw*b]
__global__ void __launch_bounds__(256) k_weight_relayout_tc(const float* __restrict__ W, float* __restrict__ A2, int Co,
                                                            int Ci, int KH, int KW, int TA, int TB, int kh0, int dkh, int kw0,
                                                            int dkw, int mode) {
    const int Mr = mode ? Ci : Co, Cr = mode ? Co : Ci;
    const long n = (long)Mr * TA * TB * Cr;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int c = (int)(i % Cr);
        const int b = (int)((i / Cr) % TB);
        const int a = (int)((i / ((long)Cr * TB)) % TA);
        const int m = (int)(i / ((long)Cr * TB * TA));
        const int co = mode ? c : m, ci = mode ? m : c;
        A2[i] = W[(((long)co * Ci + ci) * KH + kh0 + dkh * a) * KW + kw0 + dkw * b];
    }
}

inline int ew_blocks(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

template <int WAVES_M, int WAVES_N, int WM, int WN, int BKC>
void launch_cfg(const FastGemmArgs& a, int splits, hipStream_t st) {
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    const long Np = (long)a.Nb * a.NY * a.NX;
    const int gx = fd_cdiv(Np, BN), gy = fd_cdiv(a.M, BM);
    FastGemmArgs g = a;
    g.xcd_swizzle = (gx % 8 == 0 && gx >= 16) ? 1 : 0;
    const size_t lds = sizeof(float) * 2 * BKC * ((BM + 1) + BN);
    auto kern = k_conv_fast<WAVES_M, WAVES_N, WM, WN, BKC>;
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {   // allow > 64 KiB of dynamic LDS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    hipLaunchKernelGGL(kern, dim3(gx, gy, splits), dim3(64 * WAVES_M * WAVES_N), lds, st, g);
}

}  // namespace

namespace {
// Tile / split-K choice.  Workgroups resident on one CU share its four matrix pipes, so a launch takes
// ~ceil(blocks / 256) block-times: pick the configuration whose block count fills the 256 CUs most evenly
// (measured: 360 blocks of 128x128 leave 30 % of the chip idle, 720 blocks of 64x128 do not).
struct FastChoice { int cfg; int splits; };   // cfg 0: 128x128 (FD_CONV_FORCE only), 1: 64x128, 2: 32x256
FastChoice choose_config(const FastGemmArgs& a) {
    if (fd_tun().force_cfg >= 0) {                        // tuning aid (scripts/conv_cfg_sweep.py): fd_tuning.force_cfg / force_splits
        const int c = fd_tun().force_cfg, sp = fd_tun().force_splits;
        const bool ok_c = (c == 0 && a.M > 64) || (c == 1 && a.M > 32) || c == 2;
        const bool can = a.osy == 1 && a.osx == 1 && a.slab_stride > 0;
        if (ok_c) return {c, can ? sp : 1};
    }
    const long Np = (long)a.Nb * a.NY * a.NX;
    const int bkc = (a.C % 32 == 0) ? 32 : 16;
    const int nchunk = a.T * (a.C / bkc);
    const bool can_split = a.osy == 1 && a.osx == 1 && a.slab_stride > 0;
    const int bm[3] = {128, 64, 32}, bn[3] = {128, 128, 256};
    // Measured rule (scripts/conv_cfg_sweep.py, every ResNet / decoder shape at batch 12 and 24): the 64x128 tile (32x256 for
    // <= 32 output channels) with split-K chosen so that tiles x splits comes closest to, without exceeding, the 768
    // workgroups the chip holds (3 per CU) wins or ties everywhere: 720 = 720x1 = 360x2 = 180x4, 768 = 96x8 = 48x16.
    const int c = a.M > 32 ? 1 : 2;
    const long tiles = (long)fd_cdiv(Np, bn[c]) * fd_cdiv(a.M, bm[c]);
    const long fill = fd_tun().conv_target;
    int sp = 1;
    if (can_split && tiles < fill) {
        sp = (int)(fill / tiles);
        const int cap = nchunk / 4 > 0 ? (nchunk / 4 < 16 ? nchunk / 4 : 16) : 1;
        if (sp > cap) sp = cap;
        if (sp < 1) sp = 1;
    }
    return {c, sp};
}
}  // namespace

long fast_splitk_slab_floats(const FastGemmArgs& a, int* splits_out) {
    FastGemmArgs probe = a;
    if (probe.slab_stride <= 0) probe.slab_stride = probe.out_total > 0 ? probe.out_total : 1;   // sizing query
    const FastChoice ch = choose_config(probe);
    if (splits_out) *splits_out = ch.splits;
    return ch.splits > 1 ? (long)ch.splits * a.out_total : 0;
}

namespace {
template <int WAVES_M, int WAVES_N, int WM, int WN, int BKC>
void launch_grp(const FastGemmArgs& a, const FastGemmGroup& q, hipStream_t st) {
    constexpr int BM = WAVES_M * 32 * WM, BN = WAVES_N * 32 * WN;
    FastGemmGroup grp = q;
    int gx = 0;
    for (int j = 0; j < q.n; ++j) { grp.first_bx[j] = gx; gx += fd_cdiv((long)a.Nb * q.NY[j] * q.NX[j], BN); }
    grp.first_bx[q.n] = gx;
    const size_t lds = sizeof(float) * 2 * BKC * ((BM + 1) + BN);
    auto kern = k_conv_fast_grp<WAVES_M, WAVES_N, WM, WN, BKC>;
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    hipLaunchKernelGGL(kern, dim3(gx, fd_cdiv(a.M, BM), 1), dim3(64 * WAVES_M * WAVES_N), lds, st, a, grp);
}
}  // namespace

// `a` holds what the problems share; q their differences (q.first_bx is filled here).  Same tile configuration rule as
// fast_gemm_launch (64x128, or 32x256 for <= 32 output channels); no split-K.
int fast_gemm_group_launch(const FastGemmArgs& a, const FastGemmGroup& q, hipStream_t st) {
    if ((double)a.Nb * a.C * a.Hi * a.Wi * 4.0 >= 2147483648.0) { fd_set_error("conv: tensor exceeds the 2 GiB addressing range of the fast path"); return -1; }
    for (int j = 0; j < q.n; ++j)
        if ((double)a.M * q.K[j] * 4.0 >= 2147483648.0) { fd_set_error("conv: weights exceed the 2 GiB addressing range of the fast path"); return -1; }
    const bool b32 = a.C % 32 == 0;
    long wgs128 = 0;                                         // workgroups of the 64 x 128 tiling
    for (int j = 0; j < q.n; ++j) wgs128 += fd_cdiv((long)a.Nb * q.NY[j] * q.NX[j], 128) * fd_cdiv(a.M, 64);
    if (a.M > 32 && b32 && wgs128 < (long)fd_tun().grp_tile64_below) launch_grp<2, 2, 1, 1, 32>(a, q, st);
    else if (a.M > 32) { if (b32) launch_grp<2, 2, 1, 2, 32>(a, q, st); else launch_grp<2, 2, 1, 2, 16>(a, q, st); }
    else { if (b32) launch_grp<1, 4, 1, 2, 32>(a, q, st); else launch_grp<1, 4, 1, 2, 16>(a, q, st); }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_conv_fast_grp launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int fast_gemm_launch(const FastGemmArgs& a, hipStream_t st) {
    if ((double)a.Nb * a.C * a.Hi * a.Wi * 4.0 >= 2147483648.0 || (double)a.M * a.K * 4.0 >= 2147483648.0) {
        fd_set_error("conv: tensor exceeds the 2 GiB addressing range of the fast path"); return -1;
    }
    const FastChoice ch = choose_config(a);
    const int splits = ch.splits;
    if (splits > 1 && !a.slabs) { fd_set_error("conv: split-K workspace missing"); return -1; }
    const bool b32 = a.C % 32 == 0;
    if (ch.cfg == 0) {
        if (b32) launch_cfg<2, 2, 2, 2, 32>(a, splits, st); else launch_cfg<2, 2, 2, 2, 16>(a, splits, st);
    } else if (ch.cfg == 1) {
        if (b32) launch_cfg<2, 2, 1, 2, 32>(a, splits, st); else launch_cfg<2, 2, 1, 2, 16>(a, splits, st);
    } else {
        if (b32) launch_cfg<1, 4, 1, 2, 32>(a, splits, st); else launch_cfg<1, 4, 1, 2, 16>(a, splits, st);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_conv_fast launch failed: %s", hipGetErrorString(e)); return (int)e; }
    if (splits > 1)
        return fast_splitk_finish_launch(a.slabs, a.Y, a.bias, a.out_total, a.slab_stride, splits, a.out_cs, a.M, a.act, st, a.add);
    return 0;
}

int fast_wgrad_finish_launch(const float* slabs, float* gw, int M, int C, int T, int splits, int accumulate, hipStream_t st) {
#ifdef FD_ABLATE_NO_FINISH      // timing experiment only (wrong results): what the step would gain if the slab reductions cost nothing
    return 0;
#endif
    if ((T == 9 && C % 32 == 0) || T == 12) {
        // enough (m, c block) pairs for every CU, several output channels per workgroup beyond that
        const int cb = (C + 31) / 32;
        int mb = (int)(((long)M * cb) / 1024);
        mb = mb < 1 ? 1 : (mb > 8 ? 8 : mb);
        const dim3 grid(cb, (M + mb - 1) / mb);
        if (T == 9) hipLaunchKernelGGL(k_wgrad_finish9<9>, grid, dim3(384), 0, st, slabs, gw, M, C, splits, accumulate, mb);
        else hipLaunchKernelGGL(k_wgrad_finish9<12>, grid, dim3(384), 0, st, slabs, gw, M, C, splits, accumulate, mb);
        hipError_t e9 = hipGetLastError();
        if (e9 != hipSuccess) { fd_set_error("k_wgrad_finish9 launch failed: %s", hipGetErrorString(e9)); return (int)e9; }
        return 0;
    }
    hipLaunchKernelGGL(k_wgrad_finish, dim3(ew_blocks((long)M * C * T)), dim3(256), 0, st, slabs, gw, M, C, T, splits, accumulate);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_wgrad_finish launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int fast_splitk_finish_launch(const float* slabs, float* Y, const float* bias, long total, long slab_stride, int splits, long out_cs,
                              int M, int act, hipStream_t st, const float* add) {
#ifdef FD_ABLATE_NO_FINISH
    return 0;
#endif
    hipLaunchKernelGGL(k_splitk_finish, dim3(ew_blocks(total)), dim3(256), 0, st, slabs, Y, bias, total, slab_stride, splits, out_cs, M,
                       act, add);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_splitk_finish launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int fast_weight_relayout(const float* W, float* A2, int Co, int Ci, int KH, int KW, int TA, int TB, int kh0, int dkh,
                         int kw0, int dkw, int mode, hipStream_t st) {
    const long n = (long)Co * Ci * TA * TB;
    hipLaunchKernelGGL(k_weight_relayout_tc, dim3(ew_blocks(n)), dim3(256), 0, st, W, A2, Co, Ci, KH, KW, TA, TB, kh0, dkh,
                       kw0, dkw, mode);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_weight_relayout_tc launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

int fast_wgrad_splits(int M, int C, int T, long Np) {
    // Workgroups co-resident on a CU share its matrix pipes, so a launch finishes when the fullest CU does: 513 workgroups
    // on 256 CUs (one CU with 3) take 1.5x the time of 512.  Aim at 3 per CU (what the 50 KB LDS tiles allow) and never
    // exceed it; measured in the training step against 256 / 512 / 1024 and against a rounds-based cost model.
    const long target = fd_tun().wgrad_target;
    const int bn = C >= 128 ? 128 : 64;
    const long tiles = (long)T * fd_cdiv(C, bn) * fd_cdiv(M, M > 32 ? 64 : 32);
    long sp = target / tiles;
    const long maxs = (Np + 511) / 512;
    if (sp > maxs) sp = maxs;
    if (sp < 1) sp = 1;
    if (sp > 64) sp = 64;
    return (int)sp;
}

int fast_wgrad_launch(const FastWgradArgs& a, float* gw, int splits, int accumulate, hipStream_t st) {
    FastWgradArgs g = a;
    const long Np = (long)a.Nb * a.NY * a.NX;
    if ((double)a.Nb * a.C * a.Hi * a.Wi * 4.0 >= 2147483648.0 || (double)a.Nb * (double)a.dy_ns * 4.0 >= 2147483648.0) {
        fd_set_error("conv wgrad: tensor exceeds the 2 GiB addressing range of the fast path"); return -1;
    }
    long pps = (Np + splits - 1) / splits;
    pps = (pps + 31) / 32 * 32;
    g.pix_per_split = pps;
    auto go = [&](auto kern, int BM, int BN) {
        const size_t lds = sizeof(float) * 2 * 32 * ((BM + 1) + (BN + 1));
        dim3 grid(a.T * fd_cdiv(a.C, BN), fd_cdiv(a.M, BM), splits);
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, g);
    };
    if (a.M <= 32) go(k_wgrad_fast<1, 4, 1, 1>, 32, 128);
    else if (a.C >= 128) go(k_wgrad_fast<2, 2, 1, 2>, 64, 128);
    else go(k_wgrad_fast<2, 2, 1, 1>, 64, 64);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_wgrad_fast launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return fast_wgrad_finish_launch(a.slabs, gw, a.M, a.C, a.T, splits, accumulate, st);
}
