// 3x3 stride-1 convolutions with the Winograd F(2,3) transform along x (1-D), on the FP32 MFMA.
//
// For an output pair (x, x+1) = (2j, 2j+1) of row y and each of the three kernel rows ky, the four inputs d0..d3 at columns
// 2j-1 .. 2j+2 of row y+ky-1 become  V = (d0-d2, d1+d2, d2-d1, d1-d3); the kernel row (g0,g1,g2) becomes
// U = (g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2) once per optimiser step; then with  M_t = sum_{c,ky} U_t V_t  (4 independent GEMMs, M = Cout,
// N = pixel PAIRS, K = 3*Cin)  the two outputs are  M0+M1+M2  and  M1-M2-M3.  6 multiplies per output instead of 9: the matrix
// pipe - the resource that bounds the training step - does 1.5x less work for the same convolution; the transform
// arithmetic (4 adds per 4 loaded values, 4 adds per 2 outputs) rides in the loader / epilogue.  Coefficients are +-1 and 1/2, so
// the fp32 rounding error stays within a few ulps of the direct sum (tests: 1e-5 of the output scale).
//
// Kernel shape: 256 threads = 4 waves, wave t owns component t and a 64 (channels) x 64 (pairs) accumulator block = 2x2 MFMA
// 32x32x2 tiles (4 MFMAs per 4 LDS operand reads); a chunk is 16 input channels of one kernel row; LDS double-buffered
// (66 KB -> 2 workgroups per CU); the four component blocks meet in LDS for the output transform; split-K over the (ky, channel)
// chunks writes partial OUTPUTS to the usual slabs (the transform is linear), finished by k_splitk_finish.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x2 fd_ldg64(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float wino_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : expm1f(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    if (act == 4) return tanhf(v);
    return v;
}

constexpr int WBM = 64, WBN = 64, WBKC = 16, WNT = 256;
constexpr int LDU = WBM + 1, LDV = WBN, LDM = WBN + 1;
constexpr int W_BUF_FLOATS = 4 * WBKC * (LDU + LDV);          // one operand buffer (all four components)
// double-buffered operands: 66 KB -> 2 workgroups per CU.  (A single-buffered variant - 33 KB, 4 per CU, two barriers per chunk - and a
// one-chunk-deep register pipeline both measured the same; an 8-channel-chunk variant - 33 KB, 3 per CU - was 2-5 % faster alone
// and 2 % slower inside the training step, where its extra resident waves take CUs from the other streams' kernels.)
constexpr int W_LDS_FLOATS = (2 * W_BUF_FLOATS > 4 * WBM * LDM) ? 2 * W_BUF_FLOATS : 4 * WBM * LDM;

// U[t][m][ky][c] from W[m][c][ky][kx] (forward) or, for the data gradient (flip = 1: a conv over dY with the spatially flipped,
// channel-transposed kernel), from W[c][m][2-ky][2-kx].
__global__ void k_wino_weight(const float* __restrict__ w, float* __restrict__ U, int M, int C, int flip) {
    const long n = (long)M * 3 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ky = (int)((i / C) % 3);
        const int m = (int)(i / (3L * C));
        float g0, g1, g2;
        if (!flip) {
            const float* p = w + (((long)m * C + c) * 3 + ky) * 3;
            g0 = p[0]; g1 = p[1]; g2 = p[2];
        } else {
            const float* p = w + (((long)c * M + m) * 3 + (2 - ky)) * 3;
            g0 = p[2]; g1 = p[1]; g2 = p[0];
        }
        U[i] = g0;
        U[n + i] = 0.5f * (g0 + g1 + g2);
        U[2 * n + i] = 0.5f * (g0 - g1 + g2);
        U[3 * n + i] = g2;
    }
}

struct WinoArgs {
    const float* U; const float* X; float* Y; const float* bias; float* slabs;
    const float* add;    // optional, laid out like Y: Y = act(conv + bias) + add
    long slab_stride;
    int M, C, Nb, H, W;
    int pad_mode, act;
    int xcd_swizzle;     // consecutive pixel tiles (vertical neighbours share input rows) go to the same XCD / L2
};

__global__ void __launch_bounds__(WNT) k_conv_wino(WinoArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, comp = tid >> 6;       // wave index = Winograd component
    const int W2 = g.W >> 1;
    const long plane2 = (long)g.H * W2, Np = (long)g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    const int m0 = blockIdx.y * WBM;
    int bx = blockIdx.x;
    if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const long p0 = (long)bx * WBN;
    const int cpt = g.C / WBKC, nchunk_all = 3 * cpt;
    const int nsplit = (int)gridDim.z, zs = (int)blockIdx.z;
    const int per_split = (nchunk_all + nsplit - 1) / nsplit;
    const int ch_lo = zs * per_split;
    const int ch_hi = ch_lo + per_split < nchunk_all ? ch_lo + per_split : nchunk_all;

    // ---- activation loader: this thread always fetches pair jn of the tile, channel rows kr + 4 i
    const int jn = tid & 63;
    const int kr = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long pg = p0 + jn;
    const bool pvalid = pg < Np;
    int y0, j0;
    unsigned nbase;
    {
        const long pp = pvalid ? pg : 0;
        const int n = (int)(pp / plane2);
        const int rem = (int)(pp - (long)n * plane2);
        y0 = rem / W2; j0 = rem - y0 * W2;
        nbase = (unsigned)n * (unsigned)g.C * hw;
    }
    const bool refl = g.pad_mode == 1;
    const bool left_edge = j0 == 0, right_edge = 2 * j0 + 2 >= g.W;
    // ---- weight loader: float4 a4 (of the chunk's 16 channels) of row ar, for each component
    const int a4 = tid & 3, ar = tid >> 2;
    int mrow = m0 + ar;
    mrow = mrow < g.M ? mrow : g.M - 1;                                  // rows >= M are never stored
    const unsigned u_comp = 4u * (unsigned)g.M * 3u * (unsigned)g.C;     // bytes between components
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U), rsX = fd_make_rsrc(g.X);

    float4 ru[4];
    f32x2 rmid[4];
    float rl[4], rr[4];
    unsigned u_off = FD_OOB, mid_off = FD_OOB, l_off = FD_OOB, r_off = FD_OOB;
    const unsigned c_step = 4u * 4u * hw;                                // 4 channel rows further
    int pc_ky, pc_c0;
    { pc_ky = ch_lo / cpt; pc_c0 = (ch_lo - pc_ky * cpt) * WBKC; }
    auto prep_chunk = [&](bool live) __attribute__((always_inline)) {
        u_off = live ? 4u * (((unsigned)mrow * 3u + (unsigned)pc_ky) * (unsigned)g.C + (unsigned)pc_c0 + 4u * a4) : FD_OOB;
        int r = y0 + pc_ky - 1;
        const bool inb = (unsigned)r < (unsigned)g.H;
        if (refl) r = r < 0 ? -r : (r >= g.H ? 2 * g.H - 2 - r : r);
        const bool ok = pvalid & live & (refl | inb);
        const unsigned base = 4u * (nbase + (unsigned)(pc_c0 + kr) * hw + (unsigned)(r * g.W + 2 * j0));
        mid_off = ok ? base : FD_OOB;
        l_off = (ok & !left_edge) ? base - 4u : FD_OOB;
        r_off = (ok & !right_edge) ? base + 8u : FD_OOB;
        pc_c0 += WBKC;
        if (pc_c0 >= g.C) { pc_c0 = 0; ++pc_ky; }
    };
    auto load_u = [&](int t) __attribute__((always_inline)) { ru[t] = fd_ldg128(rsU, u_off + (unsigned)t * u_comp); };   // FD_OOB + (< 2^31) stays out of range
    auto load_v = [&](int i) __attribute__((always_inline)) {
        const unsigned s = (unsigned)i * c_step;
        rmid[i] = fd_ldg64(rsX, mid_off + s);                             // an FD_OOB base + (offset < 2^31) is still >= 2^31: reads 0
        rl[i] = fd_ldg32(rsX, l_off + s);
        rr[i] = fd_ldg32(rsX, r_off + s);
    };
    auto store_u = [&](int buf, int t) __attribute__((always_inline)) {
        float* q = smem + buf * W_BUF_FLOATS + t * WBKC * LDU + (4 * a4) * LDU + ar;
        q[0] = ru[t].x; q[LDU] = ru[t].y; q[2 * LDU] = ru[t].z; q[3 * LDU] = ru[t].w;
    };
    auto store_v = [&](int buf, int i) __attribute__((always_inline)) {
        const float d1 = rmid[i].x, d2 = rmid[i].y;
        const float d0 = (refl & left_edge) ? d2 : rl[i];               // reflect: column -1 is column 1, column W is column W-2
        const float d3 = (refl & right_edge) ? d1 : rr[i];
        float* q = smem + buf * W_BUF_FLOATS + 4 * WBKC * LDU + (kr + 4 * i) * LDV + jn;
        q[0] = d0 - d2;
        q[WBKC * LDV] = d1 + d2;
        q[2 * WBKC * LDV] = d2 - d1;
        q[3 * WBKC * LDV] = d1 - d3;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NK = WBKC / 2;       // 8 MFMA k-steps per chunk
    constexpr int LS = NK / 2;         // loads in the first 4 k-steps, LDS stores in the last 4
    const int arow = lane >> 5, acol = lane & 31;
    if (ch_lo < ch_hi) {
        prep_chunk(true);
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) load_v(i);
#pragma unroll
        for (int t = 0; t < 4; ++t) store_u(0, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_v(0, i);
        __syncthreads();
        for (int ch = ch_lo; ch < ch_hi; ++ch) {
            const int cur = (ch - ch_lo) & 1;
            prep_chunk(ch + 1 < ch_hi);
            const float* pa = smem + cur * W_BUF_FLOATS + comp * WBKC * LDU + arow * LDU + acol;
            const float* pb = smem + cur * W_BUF_FLOATS + 4 * WBKC * LDU + comp * WBKC * LDV + arow * LDV + acol;
            float av[2][2], bv[2][2];
            av[0][0] = pa[0]; av[0][1] = pa[32]; bv[0][0] = pb[0]; bv[0][1] = pb[32];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk + 1 < NK) {
                    av[nb][0] = pa[(kk + 1) * 2 * LDU]; av[nb][1] = pa[(kk + 1) * 2 * LDU + 32];
                    bv[nb][0] = pb[(kk + 1) * 2 * LDV]; bv[nb][1] = pb[(kk + 1) * 2 * LDV + 32];
                }
                if (kk < LS) { load_u(kk); load_v(kk); }
                else { store_u(cur ^ 1, kk - LS); store_v(cur ^ 1, kk - LS); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][i], bv[cb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    }

    // ---- output transform through LDS: sM[t][m][pair]
    float* sM = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                sM[(comp * WBM + m) * LDM + j * 32 + acol] = acc[i][j][r];
            }
    __syncthreads();
    if (!pvalid) return;
    const bool final_pass = nsplit == 1;
    float* Y = final_pass ? g.Y : g.slabs + (size_t)zs * g.slab_stride;
    {
        const int n = (int)(pg / plane2);
        const long po = ((long)n * g.M) * hw + (long)y0 * g.W + 2 * j0;
        float* yo = Y + po;
        const float* ao = (final_pass && g.add) ? g.add + po : nullptr;
#pragma unroll 4
        for (int t = 0; t < 16; ++t) {
            const int ml = kr + 4 * t, m = m0 + ml;
            if (m >= g.M) break;
            const float M0 = sM[(0 * WBM + ml) * LDM + jn], M1 = sM[(1 * WBM + ml) * LDM + jn];
            const float M2 = sM[(2 * WBM + ml) * LDM + jn], M3 = sM[(3 * WBM + ml) * LDM + jn];
            f32x2 o;
            o.x = (M0 + M1) + M2;
            o.y = (M1 - M2) - M3;
            if (final_pass) {
                const float b = g.bias ? g.bias[m] : 0.f;
                o.x = wino_act(o.x + b, g.act); o.y = wino_act(o.y + b, g.act);
                if (ao) {
                    const f32x2 a2 = *reinterpret_cast<const f32x2*>(ao + (long)m * hw);
                    o.x += a2.x; o.y += a2.y;
                }
            }
            *reinterpret_cast<f32x2*>(yo + (long)m * hw) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
// dW[m][c][ky][kx] = sum over pixels dY[m][y][x] * X[c][y+ky-1][x+kx-1].  Per pixel pair (dy0, dy1) and the same four inputs
// d0..d3 as the forward, the transposed F(2,3) algorithm needs 4 products instead of 6:
//   P = (dy0, dy0+dy1, dy0-dy1, dy1),  Q = (d0-d2, d1+d2, d2-d1, d1-d3)  (Q is the forward's input transform),
//   M_t = sum_pairs P_t Q_t   (4 GEMMs, M = Cout, N = Cin, K = pixel pairs),
//   dW[kx=0] = M0 + (M1+M2)/2,  dW[1] = (M1-M2)/2,  dW[2] = (M1+M2)/2 - M3.
// One workgroup = (64 output channels) x (64 input channels) x one kernel row ky x a slice of the pairs; wave t owns component t.
// Slices write [split][m][ky*3+kx][c] slabs, reduced in fixed order by k_wgrad_finish (deterministic).
struct WinoWgradArgs {
    const float* dY; const float* X; float* slabs;
    int M, C, Nb, H, W;
    int pad_mode;
    long pairs_per_split;
    int slice_major;     // 1: grid x = pixel slice (XCD-aligned), z = (ky, c tile); 0: x = (ky, c tile), z = slice
};
constexpr int WGP = 16;                                          // pairs per chunk (GEMM-K 16 -> 8 MFMA k-steps)
constexpr int WG_BUF_FLOATS = 4 * WGP * (LDU + LDU);             // A: [4][16][65], B: [4][16][65]
constexpr int WG_LDS_FLOATS = (2 * WG_BUF_FLOATS > 4 * WBM * LDM) ? 2 * WG_BUF_FLOATS : 4 * WBM * LDM;

__global__ void __launch_bounds__(WNT) k_wgrad_wino(WinoWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, comp = tid >> 6;
    const int W2 = g.W >> 1;
    const long plane2 = (long)g.H * W2, Np = (long)g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    // grid (default): x = (kernel row, input-channel tile), y = output-channel tile, z = pixel slice.  The alternative
    // (slice_major: x = slice with the slice count a multiple of 8, so that all workgroups of a slice share an XCD / L2) measured
    // slightly slower in the training step and is kept as a tuning switch (FD_WINO_WGRAD_MAP=1).
    const int ctiles = (g.C + WBN - 1) / WBN;
    const int bt = g.slice_major ? blockIdx.z : blockIdx.x, bs = g.slice_major ? blockIdx.x : blockIdx.z;
    const int ky = bt / ctiles, c0 = (bt - ky * ctiles) * WBN;
    const int m0 = blockIdx.y * WBM;
    const long pp_lo = (long)bs * g.pairs_per_split;
    const long pp_hi = pp_lo + g.pairs_per_split < Np ? pp_lo + g.pairs_per_split : Np;
    const int nchunk = pp_hi > pp_lo ? (int)((pp_hi - pp_lo + WGP - 1) / WGP) : 0;

    // loader: pair p of the chunk, rows rw + 16 i (dY rows = output channels, X rows = input channels)
    const int p = tid & 15, rw = tid >> 4;
    unsigned a_row[4], b_row[4];                                 // element offsets of the 4 channel rows (clamped: never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + rw + 16 * i; m = m < g.M ? m : g.M - 1;
        int c = c0 + rw + 16 * i; c = c < g.C ? c : g.C - 1;
        a_row[i] = (unsigned)m * hw; b_row[i] = (unsigned)c * hw;
    }
    const bool refl = g.pad_mode == 1;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.dY), rsX = fd_make_rsrc(g.X);
    f32x2 ra[4], rmid[4];
    float rl[4], rr[4];
    unsigned a_off = FD_OOB, mid_off = FD_OOB, l_off = FD_OOB, r_off = FD_OOB;
    bool e_left = false, e_right = false;
    long pc = pp_lo;                                             // first pair of the chunk being prepared
    auto prep_chunk = [&](bool live) __attribute__((always_inline)) {
        const long pg = pc + p;
        const bool ok = live & (pg < pp_hi);
        const long pq = ok ? pg : 0;
        const int n = (int)(pq / plane2);
        const int rem = (int)(pq - (long)n * plane2);
        const int y = rem / W2, j = rem - y * W2;
        a_off = ok ? 4u * ((unsigned)n * (unsigned)g.M * hw + (unsigned)(y * g.W + 2 * j)) : FD_OOB;
        int r = y + ky - 1;
        const bool inb = (unsigned)r < (unsigned)g.H;
        if (refl) r = r < 0 ? -r : (r >= g.H ? 2 * g.H - 2 - r : r);
        const bool okb = ok & (refl | inb);
        const unsigned base = 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)(r * g.W + 2 * j));
        e_left = j == 0; e_right = 2 * j + 2 >= g.W;
        mid_off = okb ? base : FD_OOB;
        l_off = (okb & !e_left) ? base - 4u : FD_OOB;
        r_off = (okb & !e_right) ? base + 8u : FD_OOB;
        pc += WGP;
    };
    // the edge flags belong to the chunk whose registers are in flight: latch them with the loads
    bool s_left = false, s_right = false;
    auto load_row = [&](int i) __attribute__((always_inline)) {
        ra[i] = fd_ldg64(rsY, a_off + 4u * a_row[i]);                     // FD_OOB + (< 2^31) stays out of range
        rmid[i] = fd_ldg64(rsX, mid_off + 4u * b_row[i]);
        rl[i] = fd_ldg32(rsX, l_off + 4u * b_row[i]);
        rr[i] = fd_ldg32(rsX, r_off + 4u * b_row[i]);
    };
    auto store_row = [&](int buf, int i) __attribute__((always_inline)) {
        float* qa = smem + buf * WG_BUF_FLOATS + p * LDU + rw + 16 * i;
        const float y0 = ra[i].x, y1 = ra[i].y;
        qa[0] = y0; qa[WGP * LDU] = y0 + y1; qa[2 * WGP * LDU] = y0 - y1; qa[3 * WGP * LDU] = y1;
        const float d1 = rmid[i].x, d2 = rmid[i].y;
        const float d0 = (refl & s_left) ? d2 : rl[i];
        const float d3 = (refl & s_right) ? d1 : rr[i];
        float* qb = smem + buf * WG_BUF_FLOATS + 4 * WGP * LDU + p * LDU + rw + 16 * i;
        qb[0] = d0 - d2; qb[WGP * LDU] = d1 + d2; qb[2 * WGP * LDU] = d2 - d1; qb[3 * WGP * LDU] = d1 - d3;
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int NK = WGP / 2, LS = NK / 2;
    const int arow = lane >> 5, acol = lane & 31;
    if (nchunk > 0) {
        prep_chunk(true);
        s_left = e_left; s_right = e_right;
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(i);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_row(0, i);
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            prep_chunk(ch + 1 < nchunk);
            s_left = e_left; s_right = e_right;
            const float* pa = smem + cur * WG_BUF_FLOATS + comp * WGP * LDU + arow * LDU + acol;
            const float* pb = pa + 4 * WGP * LDU;
            float av[2][2], bv[2][2];
            av[0][0] = pa[0]; av[0][1] = pa[32]; bv[0][0] = pb[0]; bv[0][1] = pb[32];
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                if (kk + 1 < NK) {
                    av[nb][0] = pa[(kk + 1) * 2 * LDU]; av[nb][1] = pa[(kk + 1) * 2 * LDU + 32];
                    bv[nb][0] = pb[(kk + 1) * 2 * LDU]; bv[nb][1] = pb[(kk + 1) * 2 * LDU + 32];
                }
                if (kk < LS) load_row(kk);
                else store_row(cur ^ 1, kk - LS);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][i], bv[cb][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        }
    }

    // ---- output transform through LDS: sM[t][m][c] -> slab[z][m][ky*3 + kx][c]
    float* sM = smem;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * arow;
                sM[(comp * WBM + m) * LDM + j * 32 + acol] = acc[i][j][r];
            }
    __syncthreads();
    const int cl = tid & 63, c = c0 + cl;
    if (c >= g.C) return;
    float* slab = g.slabs + (size_t)bs * ((size_t)g.M * 9 * g.C);
#pragma unroll 4
    for (int t = 0; t < 16; ++t) {
        const int ml = (tid >> 6) + 4 * t, m = m0 + ml;
        if (m >= g.M) break;
        const float M0 = sM[(0 * WBM + ml) * LDM + cl], M1 = sM[(1 * WBM + ml) * LDM + cl];
        const float M2 = sM[(2 * WBM + ml) * LDM + cl], M3 = sM[(3 * WBM + ml) * LDM + cl];
        const float h = 0.5f * (M1 + M2);
        float* o = slab + ((size_t)m * 9 + ky * 3) * g.C + c;
        o[0] = M0 + h;
        o[g.C] = 0.5f * (M1 - M2);
        o[2 * (size_t)g.C] = h - M3;
    }
}

inline int wino_splits(const fd_conv_desc* d, int M, int C) {
    const long tiles = (long)fd_cdiv((long)d->N * d->H * (d->W / 2), WBN) * fd_cdiv(M, WBM);
    const int nchunk = 3 * (C / WBKC);
    int sp = 1;
    static long target = 0;
    if (!target) { const char* e = getenv("FD_WINO_TARGET"); target = e ? atol(e) : 384; }     // alone on the GPU 768 is best; inside the step 256-384 (less slab traffic)
    if (tiles < target) {
        sp = (int)(target / tiles);
        const int cap = nchunk / 3 > 0 ? (nchunk / 3 < 16 ? nchunk / 3 : 16) : 1;
        if (sp > cap) sp = cap;
        if (sp < 1) sp = 1;
    }
    return sp;
}

}  // namespace

bool wino_fwd_ok(const fd_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 16 == 0 && d->W % 2 == 0 && !d->in_norm &&
           (long)d->Cout * 3 * d->Cin * 4 * 4 < 2147483648L;
}
long wino_wt_floats(int M, int C) { return 4L * M * 3 * C; }
long wino_ws_floats(const fd_conv_desc* d) {
    const int sp = wino_splits(d, d->Cout, d->Cin);
    return sp > 1 ? (long)sp * d->N * d->Cout * d->H * d->W : 0;
}
int wino_weight_launch(const float* w, float* U, int M, int C, int flip, hipStream_t st) {
    const long n = (long)M * 3 * C;
    hipLaunchKernelGGL(k_wino_weight, dim3(fd_cdiv(n, 256) > 4096 ? 4096 : fd_cdiv(n, 256)), dim3(256), 0, st, w, U, M, C, flip);
    FD_LAUNCH_CHECK("wino weight transform");
    return 0;
}
// y = act(conv3x3(x; U) + bias); d describes the convolution being computed (for a data gradient: Cin / Cout already swapped).
int wino_conv_launch(const fd_conv_desc* d, const float* x, const float* U, const float* bias, float* y, float* ws, hipStream_t st,
                     const float* add) {
    WinoArgs g = {};
    g.U = U; g.X = x; g.Y = y; g.bias = bias; g.slabs = ws; g.add = add;
    g.M = d->Cout; g.C = d->Cin; g.Nb = d->N; g.H = d->H; g.W = d->W;
    g.pad_mode = d->pad_mode; g.act = d->act;
    const long out_total = (long)d->N * d->Cout * d->H * d->W;
    g.slab_stride = out_total;
    const int sp = wino_splits(d, d->Cout, d->Cin);
    if (sp > 1 && !ws) { fd_set_error("wino conv: split-K workspace missing"); return -1; }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int gx = fd_cdiv((long)d->N * d->H * (d->W / 2), WBN), gy = fd_cdiv(d->Cout, WBM);
    g.xcd_swizzle = (gx % 8 == 0 && gx >= 16) ? 1 : 0;
    hipLaunchKernelGGL(k_conv_wino, dim3(gx, gy, sp), dim3(WNT), sizeof(float) * W_LDS_FLOATS, st, g);
    FD_LAUNCH_CHECK("k_conv_wino");
    if (sp > 1) return fast_splitk_finish_launch(ws, y, bias, out_total, out_total, sp, (long)d->H * d->W, d->Cout, d->act, st, add);
    return 0;
}

// ---- probe entry points (scripts/wino_probe.py, tests): the Winograd path on its own
extern "C" long fd_conv3x3_wino_wt_floats(const fd_conv_desc* d) { return (d && wino_fwd_ok(d)) ? wino_wt_floats(d->Cout, d->Cin) : 0; }
extern "C" long fd_conv3x3_wino_ws_floats(const fd_conv_desc* d) { return (d && wino_fwd_ok(d)) ? wino_ws_floats(d) : 0; }
extern "C" int fd_conv3x3_wino_fwd(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                                   int wt_ready, float* ws, void* stream) {
    FD_REQUIRE(d && x && w && y && wt, "fd_conv3x3_wino_fwd: NULL argument");
    FD_REQUIRE(wino_fwd_ok(d), "fd_conv3x3_wino_fwd: needs a 3x3 stride-1 pad-1 convolution with Cin %% 16 == 0 and an even width");
    hipStream_t st = (hipStream_t)stream;
    if (!wt_ready)
        if (int rc = wino_weight_launch(w, wt, d->Cout, d->Cin, 0, st)) return rc;
    return wino_conv_launch(d, x, wt, bias, y, ws, st);
}

// ---- weight gradient
bool wino_wgrad_ok(const fd_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 16 == 0 && d->Cin >= 64 && d->Cout >= 64 &&
           d->W % 2 == 0 && !d->in_norm;
}
int wino_wgrad_splits(const fd_conv_desc* d) {
    const long tiles = 3L * fd_cdiv(d->Cin, WBN) * fd_cdiv(d->Cout, WBM);
    const long Np = (long)d->N * d->H * (d->W / 2);
    static long target = 0;
    if (!target) { const char* e = getenv("FD_WINO_WGRAD_TARGET"); target = e ? atol(e) : 384; }     // in-step optimum (768: -1 %)
    long sp = target / tiles;
    const long maxs = (Np + 4 * WGP - 1) / (4 * WGP);          // at least 4 chunks per split
    if (sp > maxs) sp = maxs;
    if (sp > 512) sp = 512;
    if (sp >= 8) sp &= ~7L;                                     // XCD alignment, see k_wgrad_wino
    if (sp < 1) sp = 1;
    return (int)sp;
}
long wino_wgrad_ws_floats(const fd_conv_desc* d) { return (long)wino_wgrad_splits(d) * d->Cout * 9 * d->Cin; }
int wino_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate, hipStream_t st) {
    WinoWgradArgs g = {};
    g.dY = gy; g.X = x; g.slabs = ws;
    g.M = d->Cout; g.C = d->Cin; g.Nb = d->N; g.H = d->H; g.W = d->W; g.pad_mode = d->pad_mode;
    const int sp = wino_wgrad_splits(d);
    const long Np = (long)d->N * d->H * (d->W / 2);
    long pps = (Np + sp - 1) / sp;
    pps = (pps + WGP - 1) / WGP * WGP;
    g.pairs_per_split = pps;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    static int slice_major = -1;
    if (slice_major < 0) { const char* e = getenv("FD_WINO_WGRAD_MAP"); slice_major = e ? atoi(e) : 0; }     // measured: 450-452 (0) vs 447-449 (1) images/s
    g.slice_major = slice_major;
    const int nt = 3 * fd_cdiv(d->Cin, WBN);
    hipLaunchKernelGGL(k_wgrad_wino, slice_major ? dim3(sp, fd_cdiv(d->Cout, WBM), nt) : dim3(nt, fd_cdiv(d->Cout, WBM), sp), dim3(WNT), sizeof(float) * WG_LDS_FLOATS, st,
                       g);
    FD_LAUNCH_CHECK("k_wgrad_wino");
    return fast_wgrad_finish_launch(ws, gw, d->Cout, d->Cin, 9, sp, accumulate, st);
}

