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
#include "conv_limb.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x2 __attribute__((__vector_size__(2 * sizeof(unsigned int))));

__device__ __forceinline__ f32x2 fd_ldg64(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
// ELU / sigmoid / tanh (decoder layers): one out-of-line copy, so that the fully unrolled epilogue (32 values per lane) does not
// carry 32 inlined copies of three libm routines
__device__ __attribute__((noinline)) float wino_act_slow(float v, int act) {
    if (act == 2) return v > 0.f ? v : expm1f(v);
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    return tanhf(v);
}
__device__ __forceinline__ float wino_act(float v, int act) {
    if (act >= 2) return wino_act_slow(v, act);
    return act == 1 ? fmaxf(v, 0.f) : v;
}

#ifndef FD_WINO_ABLATE
#define FD_WINO_ABLATE 0     // timing experiments only (wrong results): 1 no main loop, 2 no output stores, 4 no epilogue at all, 8 no prologue loads,
                             // 16 loop loads out of range (no traffic), 32 no operand LDS stores in the loop, 64 loop loads from a 16 KB window
#endif
constexpr int WBM = 64, WBN = 64, WBKC = 16, WNT = 256;
constexpr int LDU = WBM + 1, LDV = WBN, LDM = WBN + 1;
// k_conv_wino keeps the activations RAW in LDS - one row of the tile's 128 pixels per channel: [0] a cell that stays 0.0,
// [3] the pixel left of the tile, [4 .. 131] the tile, [132] the pixel right of it - and applies the input transform when the
// B operands are read: 8.7 KB per chunk instead of the 16.4 KB of four transformed components (the VGPR -> LDS store path is what
// bounds the main loop, scripts/wino_ksweep.py), and 50.7 KB per workgroup = three workgroups per CU.
constexpr int LDR = 2 * WBN + 8;
constexpr int V_RAW_FLOATS = 9 * 64 * 4;                      // 16 rows x 136 = 2176 floats, rounded up to 9 wave-wide 16-byte DMAs
constexpr int W_BUF_FLOATS = 4 * WBKC * LDU + V_RAW_FLOATS;   // one operand buffer: U (four components) + raw activations
// double-buffered operands: 66 KB -> 2 workgroups per CU.  (A single-buffered variant - 33 KB, 4 per CU, two barriers per chunk - and a
// one-chunk-deep register pipeline both measured the same; an 8-channel-chunk variant - 33 KB, 3 per CU - was 2-5 % faster alone
// and 2 % slower inside the training step, where its extra resident waves take CUs from the other streams' kernels.)
constexpr int W_LDS_FLOATS = 2 * W_BUF_FLOATS;       // k_conv_wino: the double-buffered operands (its output transform stays in registers)

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

// TWOD (k_conv_wino<.., true>): U2[t][m][ri][c], the vertical transform (g_0, (g_0+g_1+g_2)/2, (g_0-g_1+g_2)/2, g_2)[ri] of the three
// kernel rows applied first, then the horizontal one: the 16 components of F(2x2, 3x3), four per row component.
__global__ void k_wino_weight2d(const float* __restrict__ w, float* __restrict__ U, int M, int C, int flip) {
    const long n = (long)M * 4 * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int ri = (int)((i / C) % 4);
        const int m = (int)(i / (4L * C));
        float g[3][3];
        const float* p = flip ? w + ((long)c * M + m) * 9 : w + ((long)m * C + c) * 9;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = flip ? p[(2 - a) * 3 + (2 - b)] : p[a * 3 + b];
        float v[3];
#pragma unroll
        for (int b = 0; b < 3; ++b)
            v[b] = ri == 0 ? g[0][b] : (ri == 3 ? g[2][b] : (ri == 1 ? 0.5f * (g[0][b] + g[1][b] + g[2][b]) : 0.5f * (g[0][b] - g[1][b] + g[2][b])));
        U[i] = v[0];
        U[n + i] = 0.5f * (v[0] + v[1] + v[2]);
        U[2 * n + i] = 0.5f * (v[0] - v[1] + v[2]);
        U[3 * n + i] = v[2];
    }
}

// The same 16 components as the split-precision image k_conv_wino2d_limb reads (conv_limb.h arithmetic): for row component ri, K-chunk
// (16 input channels), horizontal component t, limb L, K half h and output channel m one 16-byte piece of 8 bf16,
//   piece index = ((((ri * C/16 + chunk) * 4 + t) * 3 + L) * 2 + h) * M + m
// - a chunk's 24 planes of M consecutive pieces are what the kernel copies into LDS, a lane's MFMA fragment is one piece.
__host__ __device__ inline long wino_limb_piece(int ri, int chunk, int t, int L, int h, long m, long M, int cpt) {
    return ((((long)(ri * cpt + chunk) * 4 + t) * 3 + L) * 2 + h) * M + m;
}
__global__ void k_wino_weight2d_limb(const float* __restrict__ w, uint4* __restrict__ A3, int M, int C, int flip) {
    const int c8n = C >> 3, cpt = C >> 4;
    const long n = (long)M * 4 * c8n;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int m = (int)(i % M);
        const int ri = (int)((i / M) % 4);
        const int c8 = (int)(i / (4L * M));
        float u[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c8 * 8 + e;
            float g[3][3];
            const float* p = flip ? w + ((long)c * M + m) * 9 : w + ((long)m * C + c) * 9;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) g[a][b] = flip ? p[(2 - a) * 3 + (2 - b)] : p[a * 3 + b];
            float v[3];
#pragma unroll
            for (int b = 0; b < 3; ++b)
                v[b] = ri == 0 ? g[0][b] : (ri == 3 ? g[2][b] : (ri == 1 ? 0.5f * (g[0][b] + g[1][b] + g[2][b]) : 0.5f * (g[0][b] - g[1][b] + g[2][b])));
            u[0][e] = v[0]; u[1][e] = 0.5f * (v[0] + v[1] + v[2]); u[2][e] = 0.5f * (v[0] - v[1] + v[2]); u[3][e] = v[2];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 h, md, l;
            fdlimb::split8(u[t], h, md, l);
            A3[wino_limb_piece(ri, c8 >> 1, t, 0, c8 & 1, m, M, cpt)] = h;
            A3[wino_limb_piece(ri, c8 >> 1, t, 1, c8 & 1, m, M, cpt)] = md;
            A3[wino_limb_piece(ri, c8 >> 1, t, 2, c8 & 1, m, M, cpt)] = l;
        }
    }
}

struct WinoArgs {
    const float* U; const float* X; float* Y; const float* bias; float* slabs;
    const float* add;    // optional, laid out like Y: Y = act(conv + bias) + add
    long slab_stride;
    int M, C, Nb, H, W;
    int pad_mode, act;
    int xcd_swizzle;     // 1: consecutive pixel tiles (vertical neighbours share input rows) go to the same XCD / L2
                         // 2 (k_conv_wino2d): 1-D grid, all pixel tiles of a (channel tile, row component, split) on one XCD
    int gx, gy, gz;      // the logical grid of xcd_swizzle == 2
    // optional: per-channel statistics of the output for the BatchNorm that follows (fd_conv2d_fwd_stats): [Nb][M][stat_slots][2] =
    // (sum, sum of squares) over the 64 pixels of each (pixel tile, 32-pair half); needs tiles that do not straddle images
    float* stat_part;
    int stat_slots;
    int img_tiles;       // k_conv_wino2p_dma: > 0 = tiles per image of the image-aligned tiling (statistics on planes of 32 k tiles)
};

// VDMA: the raw activation rows go from global memory straight into LDS (buffer_load_dwordx4 ... lds; needs W % 4 == 0 so that a
// lane's four pixels share an image row): no staging registers, no s_waitcnt + ds_write in the MFMA stream for them - the VGPR ->
// LDS stores of the activations cost 0.9 of the 6.7 us per chunk-round of the register-staged loop (scripts/wino_ksweep.py).
// STATS: the epilogue also reduces the tile to the BatchNorm partial sums (g.stat_part != nullptr).  A template flag, not a run-time
// test: with `if (g.stat_part)` around writes into the accumulator array the compiler kept BOTH versions of every remaining
// accumulator alive and emitted two v_accvgpr_read + a v_cndmask per accumulator and ROW - 700 of the 1 430 vector instructions of
// the epilogue, 15 % of the kernel's time on the layer1 shape (profiles/round3_experiments.md).
// TWOD: F(2x2, 3x3) for the deep layers, where split-K slabs are written anyway.  The GEMM-N unit becomes a 2x2 output tile (tile
// row ty = output rows 2 ty, 2 ty + 1), and blockIdx.z carries a row COMPONENT ri = z & 3 (z >> 2: split of the input channels)
// instead of a share of the (kernel row, channel) chunks: the workgroup convolves the row combination
//   (x_r0 - x_r2,  x_r1 + x_r2,  x_r2 - x_r1,  x_r1 - x_r3)[ri]      (input rows 2 ty - 1 .. 2 ty + 2, padded like the columns)
// - formed by the loader from two row loads, register-staged - with U2[.][.][ri][.] over the input channels only (a third of the
// 1-D kernel's K for four instead of one or two z), and writes the horizontally transformed products S_ri [N][M][H/2][W] to slab z.
// k_wino2d_finish applies the vertical output transform  y[2 ty] = S0 + S1 + S2,  y[2 ty + 1] = S1 - S2 - S3  (+ bias, activation,
// residual) while it sums the slabs: 16 products per 2x2 tile instead of 24, for the slab traffic of a 2-way split.
template <bool VDMA, bool STATS, bool TWOD>
__device__ __forceinline__ void conv_wino_body(const WinoArgs& g) {
    static_assert(!TWOD || (!VDMA && !STATS), "the 2-D variant is register-staged and always writes slabs");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1;
    const int HT = TWOD ? g.H >> 1 : g.H;                                // rows of GEMM-N units (pairs / 2x2 tiles) per image
    const int plane2 = HT * W2;                                          // units per image; Nb * plane2 < 2^29 (size guard)
    const int Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, nz = gridDim.z;
    if (TWOD && g.xcd_swizzle == 2) {
        // Workgroup id L runs on XCD L % 8.  The deep layers are weight-heavy (layer4: 16.8 MB of U2 against 6 MB of activations at
        // batch 24): with the pixel tile as the fastest grid index every XCD pulled every U2 slice through its own L2 - 154 MB of
        // fetches per launch (profiles/round3_pmc_conv_wino2d_layer4.md).  Here the pixel tiles of one (channel tile, row
        // component, split) slice get ids 8 apart: one XCD, back to back, the slice's 512 KB of U2 read from HBM once.
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        bx = k % g.gx;
        const int sl = (k / g.gx) * 8 + xcd;
        by = sl % g.gy; bz = sl / g.gy; nz = g.gz;
    } else if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const int m0 = by * WBM;
    const int p0 = bx * WBN;
    const int cpt = g.C / WBKC, nchunk_all = TWOD ? cpt : 3 * cpt;
    const int zs = bz;
    const int ri = TWOD ? zs & 3 : 0;                                    // row component of this workgroup
    const int nsplit = TWOD ? nz >> 2 : nz;
    const int ks = TWOD ? zs >> 2 : zs;
    constexpr unsigned UR = TWOD ? 4u : 3u;                              // weight rows per output channel
    const int xr_a = ri == 0 ? 0 : (ri == 2 ? 2 : 1), xr_b = ri == 3 ? 3 : (ri == 2 ? 1 : 2);
    const float x_sgn = ri == 1 ? 1.f : -1.f;
    const int per_split = (nchunk_all + nsplit - 1) / nsplit;
    const int ch_lo = ks * per_split;
    const int ch_hi = ch_lo + per_split < nchunk_all ? ch_lo + per_split : nchunk_all;

    // ---- activation loader: this thread always fetches pair jn of the tile, channel rows kr + 4 i
    const int jn = lane;
    const int kr = wave;
    const int pg = p0 + jn;
    const bool pvalid = pg < Np;
    int y0, j0;
    unsigned nbase;
    {
        const int pp = pvalid ? pg : 0;
        const int n = pp / plane2;
        const int rem = pp - n * plane2;
        y0 = rem / W2; j0 = rem - y0 * W2;
        nbase = (unsigned)n * (unsigned)g.C * hw;
    }
    const bool refl = g.pad_mode == 1;
    const bool left_edge = j0 == 0, right_edge = 2 * j0 + 2 >= g.W;
    // the tile's two halo pixels per channel row are fetched by lane 0 (left of its pair) and lane 63 (right of its pair); a pair
    // at an image border has no such pixel (its reader substitutes the padding value), every other lane stays out of range
    const bool halo_l = jn == 0 && !left_edge, halo_r = jn == WBN - 1 && !right_edge;
    // ---- weight loader: float4 a4 (of the chunk's 16 channels) of row ar, for each component
    const int a4 = tid & 3, ar = tid >> 2;
    int mrow = m0 + ar;
    mrow = mrow < g.M ? mrow : g.M - 1;                                  // rows >= M are never stored
    const unsigned u_comp = 4u * (unsigned)g.M * UR * (unsigned)g.C;     // bytes between components
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U), rsX = fd_make_rsrc(g.X);

    // ---- VDMA: the raw buffer is ONE linear stream of 16 rows x 34 sixteen-byte pieces (pixels -4 .. 131 of the tile's flat pixel
    //      range, row stride 136 floats); piece L = 64 * (wave + 4 q) + lane of DMA q belongs to row L / 34, piece L % 34.
    //      Per lane and DMA, fixed for the whole tile: image row / byte offset of its four pixels inside channel 0.
    const __amdgpu_buffer_rsrc_t rsXd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    unsigned d_base[3] = {FD_OOB, FD_OOB, FD_OOB};
    int d_y[3] = {0, 0, 0};
    unsigned d_off[3] = {FD_OOB, FD_OOB, FD_OOB};
    if (VDMA) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int L = 64 * (wave + 4 * q) + lane;
            const int row = L / 34, seg = L - row * 34;
            const int F = 2 * p0 - 4 + 4 * seg;                              // flat pixel index over (image, y, x)
            const bool ok = row < WBKC && F >= 0 && F < g.Nb * (int)hw;
            const int Fc = ok ? F : 0;
            const int n = Fc / (int)hw, rem = Fc - n * (int)hw;
            d_y[q] = rem / g.W;
            d_base[q] = ok ? 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)row * hw + (unsigned)rem) : FD_OOB;
        }
    }
    float4 ru[4];
    f32x2 rmid[4], rmid2[TWOD ? 4 : 1];
    float rh[4], rh2[TWOD ? 4 : 1];
    unsigned u_off = FD_OOB, mid_off = FD_OOB, h_off = FD_OOB, mid_off2 = FD_OOB, h_off2 = FD_OOB;
    unsigned d_soff = 0u;
    const unsigned c_step = 4u * 4u * hw;                                // 4 channel rows further
    int pc_ky, pc_c0;
    { pc_ky = ch_lo / cpt; pc_c0 = (ch_lo - pc_ky * cpt) * WBKC; }
    // Offsets of the next chunk to fetch, in two branch-free halves (each small enough to hide behind one MFMA, see the k-loop)
    unsigned prep_base = 0u, prep_base2 = 0u;
    bool prep_ok = false, prep_ok2 = false;
    const int H2m2 = 2 * g.H - 2;
    auto prep_a = [&](bool live) __attribute__((always_inline)) {
        u_off = live ? 4u * (((unsigned)mrow * UR + (unsigned)(TWOD ? ri : pc_ky)) * (unsigned)g.C + (unsigned)pc_c0 + 4u * a4) : FD_OOB;
        const int r = TWOD ? 2 * y0 - 1 + xr_a : y0 + pc_ky - 1;
        const bool inb = (unsigned)r < (unsigned)g.H;
        int rr_ = r < 0 ? -r : r;
        rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
        const int ruse = refl ? rr_ : r;
        prep_ok = pvalid & live & (refl | inb);
        prep_base = 4u * (nbase + (unsigned)(pc_c0 + kr) * hw + (unsigned)(ruse * g.W + 2 * j0));
        if constexpr (TWOD) {
            const int r2 = 2 * y0 - 1 + xr_b;
            const bool inb2 = (unsigned)r2 < (unsigned)g.H;
            int rr2 = r2 < 0 ? -r2 : r2;
            rr2 = rr2 >= g.H ? H2m2 - rr2 : rr2;
            const int ruse2 = refl ? rr2 : r2;
            prep_ok2 = pvalid & live & (refl | inb2);
            prep_base2 = 4u * (nbase + (unsigned)(pc_c0 + kr) * hw + (unsigned)(ruse2 * g.W + 2 * j0));
        }
        if (VDMA) {
            d_soff = 4u * (unsigned)pc_c0 * hw;                          // wave-uniform: first channel of the chunk
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int yq = d_y[q] + pc_ky - 1;
                const bool in_q = (unsigned)yq < (unsigned)g.H;
                int yr = yq < 0 ? -yq : yq;
                yr = yr >= g.H ? H2m2 - yr : yr;
                const int dyq = (refl ? yr : yq) - d_y[q];
                d_off[q] = (live & (refl | in_q)) ? d_base[q] + (unsigned)(dyq * g.W * 4) : FD_OOB;   // FD_OOB base + anything stays out of range
            }
        }
    };
    auto prep_b = [&]() __attribute__((always_inline)) {
        mid_off = prep_ok ? prep_base : FD_OOB;
        h_off = (prep_ok & halo_l) ? prep_base - 4u : ((prep_ok & halo_r) ? prep_base + 8u : FD_OOB);
        if constexpr (TWOD) {
            mid_off2 = prep_ok2 ? prep_base2 : FD_OOB;
            h_off2 = (prep_ok2 & halo_l) ? prep_base2 - 4u : ((prep_ok2 & halo_r) ? prep_base2 + 8u : FD_OOB);
        }
        if (FD_WINO_ABLATE & 16) { u_off = mid_off = h_off = FD_OOB; d_off[0] = d_off[1] = d_off[2] = FD_OOB; }   // loads issue, no memory traffic
        pc_c0 += WBKC;
        const bool wrap = pc_c0 >= g.C;
        pc_c0 = wrap ? 0 : pc_c0;
        pc_ky += wrap ? 1 : 0;
    };
    auto load_u = [&](int t) __attribute__((always_inline)) { ru[t] = fd_ldg128(rsU, u_off + (unsigned)t * u_comp); };   // FD_OOB + (< 2^31) stays out of range
    // an FD_OOB base + (offset < 2^31) is still >= 2^31: reads 0 - vertical zero padding and pairs past the end need no select
    auto load_mid = [&](int i) __attribute__((always_inline)) {
        rmid[i] = fd_ldg64(rsX, mid_off + (unsigned)i * c_step);
        if constexpr (TWOD) rmid2[i] = fd_ldg64(rsX, mid_off2 + (unsigned)i * c_step);
    };
    auto load_h = [&](int i) __attribute__((always_inline)) {
        rh[i] = fd_ldg32(rsX, h_off + (unsigned)i * c_step);
        if constexpr (TWOD) rh2[i] = fd_ldg32(rsX, h_off2 + (unsigned)i * c_step);
    };
    auto store_u = [&](int buf, int t) __attribute__((always_inline)) {
        float* q = smem + buf * W_BUF_FLOATS + t * WBKC * LDU + (4 * a4) * LDU + ar;
        q[0] = ru[t].x; q[LDU] = ru[t].y; q[2 * LDU] = ru[t].z; q[3 * LDU] = ru[t].w;
    };
    const int v_row = 4 * WBKC * LDU + kr * LDR;                         // this thread's first channel row of the raw buffer
    const int h_col = jn == 0 ? 3 : 2 * WBN + 4;                         // where a halo lane puts its pixel
    auto store_v = [&](int buf, int i) __attribute__((always_inline)) {
        float* q = smem + buf * W_BUF_FLOATS + v_row + 4 * i * LDR;
        if constexpr (TWOD) {                                            // the row combination (exact products: a +- b)
            rmid[i].x = fmaf(x_sgn, rmid2[i].x, rmid[i].x); rmid[i].y = fmaf(x_sgn, rmid2[i].y, rmid[i].y);
            rh[i] = fmaf(x_sgn, rh2[i], rh[i]);
        }
        *reinterpret_cast<f32x2*>(q + 4 + 2 * jn) = rmid[i];
        if (jn == 0 || jn == WBN - 1) q[h_col] = rh[i];
    };
    // DMA q of this wave -> the raw rows of buffer `buf` (LDS destination = wave-uniform base + 16 bytes x lane)
    auto dma_v = [&](int buf, int q) __attribute__((always_inline)) {
        float* dst = smem + buf * W_BUF_FLOATS + 4 * WBKC * LDU + (wave + 4 * q) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXd, (__attribute__((address_space(3))) void*)dst, 16, (int)d_off[q], (int)d_soff, 0, 0);
    };

    // Wave w owns the 32 (channels) x 32 (pairs) block (w >> 1, w & 1) of the tile with ALL FOUR Winograd components: one
    // accumulator per component.  The four component products of an output therefore sit in the same lane and register, and the
    // output transform (M0 + M1 + M2, M1 - M2 - M3) is plain register arithmetic in the epilogue.  (Round 1 / 2 gave each wave ONE
    // component of the whole 64 x 64 tile - half the LDS operand reads per MFMA - and met the other components in LDS: a
    // 64 KB round trip + barrier that took ~6 us per workgroup, a fifth of a 12-chunk tile; scripts/wino_ksweep.py.)
    const int wm = wave >> 1, wn = wave & 1;
    // Columns of the raw row this lane's B operands come from: d1, d2 = the pair itself, d0 / d3 = its left / right neighbour
    // pixel - the halo cells for the tile's first / last pair - or, where the pair touches an image border, the padding value:
    // for reflection padding the mirror pixel (column -1 is column 1, column W is column W - 2), for zero padding the factor 0.
    int o12, o0, o3;
    float ml, mr;                                                        // 0.0 where zero padding replaces d0 / d3
    {
        const int jp = 32 * wn + (lane & 31);
        const int pp = p0 + jp < Np ? p0 + jp : 0;
        const int rem = pp % plane2;
        const int jj = rem % W2;
        const bool le = jj == 0, re = 2 * jj + 2 >= g.W;
        o12 = 4 + 2 * jp;
        o0 = (le && refl) ? o12 : o12 - 2;       // 8-byte cell whose .y is d0 (reflection: column -1 is column 1 = d12.y)
        o3 = (re && refl) ? o12 : o12 + 2;       // 8-byte cell whose .x is d3 (reflection: column W is column W - 2 = d12.x)
        ml = (le && !refl) ? 0.f : 1.f;
        mr = (re && !refl) ? 0.f : 1.f;
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // Main loop.  One v_mfma_f32_32x32x2_f32 occupies the SIMD's matrix pipe for 64 cycles, during which the issuing wave is free
    // to issue a handful of other instructions.  Everything that is not an MFMA is therefore cut into pieces of <= 4-5
    // instructions and placed BETWEEN the four MFMAs of a k-step (sched_barrier pins the order): the operand reads of the next
    // k-step, the staging of the following chunks and their address arithmetic.  With the same instructions in one block ahead
    // of the four MFMAs (round 2) the matrix pipe idled while that block issued: scripts/ubench/mfma_ablate2.hip measures
    // 112 -> 128 TFLOP/s for this instruction mix at two workgroups per CU on random operands (124 -> 142 on constants).
    //
    // Staging pipeline, one register set, three chunks deep: in slot i (= k-step i of the first half) of chunk ch the registers
    // of slot i - loaded one whole chunk earlier - are written to the LDS buffer of chunk ch + 1 and immediately re-loaded with
    // chunk ch + 2.  Every global load thus has a full chunk (8 k-steps, >= 2 000 cycles) to return before its s_waitcnt; with
    // load and store of the same chunk four k-steps apart (round 2) the wait stalled the wave - and the MFMAs behind it - whenever
    // the fabric was slower than that (scripts/wino_ksweep.py + FD_WINO_ABLATE: 14 % of the loop time).
    constexpr int NK = WBKC / 2;       // 8 MFMA k-steps per chunk
    constexpr int LS = NK / 2;         // staging slots: k-steps 0-3
    const int arow = lane >> 5, acol = lane & 31;
    if (ch_lo < ch_hi) {
        prep_a(true); prep_b();
        if (FD_WINO_ABLATE & 8) { u_off = mid_off = h_off = FD_OOB; }
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
        if (VDMA) {
            dma_v(0, 0); dma_v(0, 1);
            if (wave == 0) dma_v(0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { load_mid(i); load_h(i); }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) store_u(0, t);
        if (!VDMA) {
#pragma unroll
            for (int i = 0; i < 4; ++i) store_v(0, i);
        }
        prep_a(ch_lo + 1 < ch_hi); prep_b();                     // chunk ch_lo + 1
        if (!VDMA) {                                             // ... loaded now, written to LDS during chunk ch_lo
#pragma unroll
            for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
            for (int i = 0; i < 4; ++i) { load_mid(i); load_h(i); }
            prep_a(ch_lo + 2 < ch_hi); prep_b();                 // offsets of chunk ch_lo + 2, re-loaded during chunk ch_lo
        } else {
            // VDMA: chunk ch + 1 is fetched DURING chunk ch (weights: k-steps 0-3 into registers, stored in k-steps 4-7; activations:
            // three DMAs) with the offsets prepared one chunk earlier; the DMAs must have landed before anyone reads them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int ch = ch_lo; ch < ((FD_WINO_ABLATE & 1) ? ch_lo : ch_hi); ++ch) {
            const int cur = (ch - ch_lo) & 1;
            // operands of component t: A = U_t[k][32 wm + acol], B = input transform of the raw row k at this lane's pair;
            // k = 2 kk + arow
            const float* pa = smem + cur * W_BUF_FLOATS + arow * LDU + 32 * wm + acol;
            const float* pr = smem + cur * W_BUF_FLOATS + 4 * WBKC * LDU + arow * LDR;
            float av[2][4], bv[2][4];
            auto read_a = [&](int nb, int k2, int t) __attribute__((always_inline)) { av[nb][t] = pa[t * WBKC * LDU + k2 * LDU]; };
            f32x2 d12, dl, dr;                                           // three 8-byte reads (conflict-free at stride 8 over a half-wave;
            auto read_b = [&](int k2) __attribute__((always_inline)) {   // the 4-byte reads of d0 / d3 at stride 8 were 2-way bank conflicts)
                d12 = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o12);
                dl = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o0); dr = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o3);
            };
            auto xform_b = [&](int nb) __attribute__((always_inline)) {   // (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
                asm volatile("" : "+v"(dl), "+v"(dr));                   // both halves live: keeps the reads 8 bytes wide
                bv[nb][0] = fmaf(dl.y, ml, -d12.y); bv[nb][1] = d12.x + d12.y; bv[nb][2] = d12.y - d12.x; bv[nb][3] = fmaf(-dr.x, mr, d12.x);
            };
#pragma unroll
            for (int t = 0; t < 4; ++t) read_a(0, 0, t);
            read_b(0); xform_b(0);
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NK) { read_b(2 * (kk + 1)); read_a(nb, 2 * (kk + 1), 0); read_a(nb, 2 * (kk + 1), 1); }
                if (!(FD_WINO_ABLATE & (32 | 128))) {
                    if (!VDMA && kk < LS) store_u(cur ^ 1, kk);
                    if (VDMA && kk >= LS) store_u(cur ^ 1, kk - LS);
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NK) { read_a(nb, 2 * (kk + 1), 2); read_a(nb, 2 * (kk + 1), 3); }
                if (kk < LS) load_u(kk);
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][2], bv[cb][2], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!VDMA && kk < LS && !(FD_WINO_ABLATE & (32 | 256))) store_v(cur ^ 1, kk);
                if (kk + 1 < NK) xform_b(nb);
                __builtin_amdgcn_sched_barrier(0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][3], bv[cb][3], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (VDMA) {
                    if (kk < 2) dma_v(cur ^ 1, kk);
                    if (kk == 2 && wave == 0) dma_v(cur ^ 1, 2);
                    if (kk == NK - 2) prep_a(ch + 2 < ch_hi);    // every fetch of chunk ch + 1 has been issued by now
                } else {
                    if (kk < LS) { load_mid(kk); load_h(kk); }
                    if (kk == NK - 2) prep_a(ch + 3 < ch_hi);    // every load of chunk ch + 2 has been issued by now
                }
                if (kk == NK - 1) prep_b();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (VDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this chunk's DMAs (into the other buffer) have landed
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }

    // ---- epilogue: output transform in registers.  C/D layout of the 32x32 MFMA: column (pair) = lane & 31,
    //      row (channel) = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    if (FD_WINO_ABLATE & 4) { if (acc[0][0] == 123.456f) g.Y[tid] = acc[1][3] + acc[2][2] + acc[3][1]; return; }
    const int po = p0 + 32 * wn + acol;                                   // this lane's output pair
    const bool final_pass = !TWOD && nsplit == 1;
    const unsigned hwo = TWOD ? (unsigned)(HT * g.W) : hw;                // plane of the tensor written: S_ri has H / 2 rows
    unsigned out_base = FD_OOB;
    if (po < Np) {
        const int n = po / plane2;
        const int rem = po - n * plane2;
        const int yy = rem / W2, jj = rem - yy * W2;
        out_base = 4u * ((unsigned)n * (unsigned)g.M * hwo + (unsigned)(yy * g.W + 2 * jj));
    }
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(final_pass ? g.Y : g.slabs + (size_t)zs * g.slab_stride);
    const __amdgpu_buffer_rsrc_t rsAdd = fd_make_rsrc(g.add ? g.add : g.Y);
    const bool has_add = final_pass && g.add;
    const int mbase = m0 + 32 * wm + 4 * arow;
    // the 16 bias values of this lane's rows: one batch of loads in front of the row loop (a load + wait per row serialised 16
    // memory latencies in the epilogue of every biased - i.e. every decoder - convolution)
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
    if (final_pass && g.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            bias_r[r] = g.bias[m < g.M ? m : g.M - 1];
        }
    }
    float s1[16], s2[16];                       // STATS: (sum, M2) of each row's two pixels
    auto rows = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;            // 0: none (compile-time), -1: g.act at run time
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            const unsigned off = (m < g.M) ? out_base + 4u * (unsigned)m * hwo : FD_OOB;      // out of range: the store is dropped
            f32x2 o;
            o.x = (acc[0][r] + acc[1][r]) + acc[2][r];
            o.y = (acc[1][r] - acc[2][r]) - acc[3][r];
            if (final_pass) {
                o.x += bias_r[r]; o.y += bias_r[r];
                if (ACT != 0) { o.x = wino_act(o.x, g.act); o.y = wino_act(o.y, g.act); }
                if (has_add) {
                    const f32x2 a2 = fd_ldg64(rsAdd, off);
                    o.x += a2.x; o.y += a2.y;
                }
            }
            if (!(FD_WINO_ABLATE & 2) || o.x == 123.456f)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsY, (int)off, 0, 0);
            if (STATS) { const float dd = o.x - o.y; s1[r] = o.x + o.y; s2[r] = 0.5f * dd * dd; }
        }
    };
    if (g.act == 0 || !final_pass) rows(std::integral_constant<int, 0>{});
    else rows(std::integral_constant<int, -1>{});
    // ---- BatchNorm statistics of the tile (fd_conv2d_fwd_stats): (sum, M2 = sum of squared deviations from the partial's OWN
    //      mean) over the 32 pairs of each half-wave for its 16 channel rows, by a transposing butterfly - after the steps 16, 8, 4,
    //      2 a lane holds ONE row's partial, the step 1 completes it: 16 cross-lane moves per statistic instead of 80, fixed order
    //      (deterministic).  Two halves of n elements each merge as M2 = M2a + M2b + (sa - sb)^2 / 2n (pairwise update of Chan
    //      et al.): no E[x^2] - E[x]^2 anywhere, so a channel whose mean is many standard deviations from zero loses nothing.
    if (STATS && final_pass) {
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int width = 8 >> step;                                   // rows kept by a lane after this step
            const int xm = 16 >> step;                                     // lane distance of the exchange
            const bool hi = (lane & xm) != 0;
            const float inv2n = 0.25f / (float)(1 << step);                // each side holds n = 2 << step pixels
#pragma unroll
            for (int j = 0; j < width; ++j) {
                const float k1 = hi ? s1[j + width] : s1[j], g1 = hi ? s1[j] : s1[j + width];
                const float k2 = hi ? s2[j + width] : s2[j], g2 = hi ? s2[j] : s2[j + width];
                const float o1 = __shfl_xor(g1, xm, 64), o2 = __shfl_xor(g2, xm, 64);
                const float df = k1 - o1;
                s1[j] = k1 + o1;
                s2[j] = fmaf(df * df, inv2n, k2 + o2);
            }
        }
        {
            const float o1 = __shfl_xor(s1[0], 1, 64), o2 = __shfl_xor(s2[0], 1, 64);
            const float df = s1[0] - o1;
            s2[0] = fmaf(df * df, 1.0f / 64.0f, s2[0] + o2);                // n = 32 per side
            s1[0] += o1;
        }
        // row held by this lane: bits (lane >> 4, lane >> 3, lane >> 2, lane >> 1) -> reg index, then the C/D layout above
        const int rr = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int m = mbase + (rr & 3) + 8 * (rr >> 2);
        if (!(lane & 1) && m < g.M) {
            const int n = p0 / plane2, tile = (p0 - n * plane2) / WBN;     // the whole tile lies in image n (launcher's guarantee)
            f32x2 v; v.x = s1[0]; v.y = s2[0];
            *reinterpret_cast<f32x2*>(g.stat_part + (((size_t)n * g.M + m) * g.stat_slots + 2 * tile + wn) * 2) = v;
        }
    }
}

template <bool VDMA, bool STATS>
__global__ void __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(3, 3))) k_conv_wino(WinoArgs g) { conv_wino_body<VDMA, STATS, false>(g); }
__global__ void __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(3, 3))) k_conv_wino2d(WinoArgs g) { conv_wino_body<false, false, true>(g); }

// y[n][m][2 ty + (0, 1)][x] = act(bias[m] + (S0 + S1 + S2,  S1 - S2 - S3)) + add, S_ri = sum over the channel splits of slab 4 ks + ri
// (fixed order => deterministic); one thread per pair of columns of a tile row.
__global__ void __launch_bounds__(256) k_wino2d_finish(const float* __restrict__ slabs, float* __restrict__ Y, const float* __restrict__ bias,
                                                       const float* __restrict__ add, unsigned total2, long slab_stride, int ksplit,
                                                       int HT, int W, int M, int act) {
    const unsigned W2 = (unsigned)W >> 1, hw2 = (unsigned)HT * W2;       // total2 = N * M * HT * W / 2 < 2^30
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total2; i += gridDim.x * 256u) {
        f32x2 s[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[r].x = 0.f; s[r].y = 0.f; }
        for (int k = 0; k < ksplit; ++k) {
            f32x2 a[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = *reinterpret_cast<const f32x2*>(slabs + (size_t)(4 * k + r) * slab_stride + 2 * (size_t)i);
#pragma unroll
            for (int r = 0; r < 4; ++r) { s[r].x += a[r].x; s[r].y += a[r].y; }
        }
        const unsigned plane = i / hw2, rem = i - plane * hw2;
        const unsigned ty = rem / W2, j = rem - ty * W2;
        const float b = bias ? bias[plane % (unsigned)M] : 0.f;
        f32x2 o0, o1;
        o0.x = (s[0].x + s[1].x) + s[2].x + b; o0.y = (s[0].y + s[1].y) + s[2].y + b;
        o1.x = (s[1].x - s[2].x) - s[3].x + b; o1.y = (s[1].y - s[2].y) - s[3].y + b;
        if (act != 0) { o0.x = wino_act(o0.x, act); o0.y = wino_act(o0.y, act); o1.x = wino_act(o1.x, act); o1.y = wino_act(o1.y, act); }
        const size_t o = ((size_t)plane * (2u * HT) + 2u * ty) * (unsigned)W + 2u * j;
        if (add) {
            const f32x2 a0 = *reinterpret_cast<const f32x2*>(add + o), a1 = *reinterpret_cast<const f32x2*>(add + o + W);
            o0.x += a0.x; o0.y += a0.y; o1.x += a1.x; o1.y += a1.y;
        }
        *reinterpret_cast<f32x2*>(Y + o) = o0;
        *reinterpret_cast<f32x2*>(Y + o + W) = o1;
    }
}

// ------------------------------------------------------------------------------------------------ F(2x2, 3x3) slabs, 128 x 64 tile
// k_conv_wino2d_m128 (round 4): k_conv_wino2d for layers with Cout % 128 == 0 (ResNet layer2 .. layer4) with TWICE the output
// channels per workgroup and wave: a wave owns 64 (channels) x 32 (2x2 tiles) = two 32 x 32 blocks per Winograd component that
// share every B operand.  The B side is the expensive one (per k-step 6 LDS reads, the row combination and the horizontal input
// transform: 8 vector instructions) and is now paid once per EIGHT matrix instructions instead of four; the activations of a pixel
// tile are fetched from L2 / HBM by half as many workgroups.  Activations go global -> LDS raw (both input rows of the row
// combination, k_conv_wino2p_dma's scheme with the row component fixed per workgroup, so the DMA offsets are computed once), weights
// register-staged.  128 accumulator registers -> two waves per SIMD; chunks of 8 input channels keep two workgroups per CU in LDS
// (2 x 26.1 KB each) at the same 32 matrix instructions per wave and barrier as the other Winograd kernels.  Same slabs, same
// k_wino2d_finish.
constexpr int M2_KC = 8, M2_BM = 128, M2_LDU = M2_BM + 1;
constexpr int M2_VRAW = 5 * 256;                                 // one raw row set: 8 rows x 34 sixteen-byte pieces in 5 wave-wide DMAs
constexpr int M2_BUF_FLOATS = 4 * M2_KC * M2_LDU + 2 * M2_VRAW;
constexpr int M2_LDS_FLOATS = 2 * M2_BUF_FLOATS;
__global__ void __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_conv_wino2d_m128(WinoArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1, HT = g.H >> 1;
    const int plane2 = HT * W2;
    const int Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, nz = gridDim.z;
    if (g.xcd_swizzle == 2) {                                    // all pixel tiles of a (channel tile, row component, split) slice on one XCD
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        bx = k % g.gx;
        const int sl = (k / g.gx) * 8 + xcd;
        by = sl % g.gy; bz = sl / g.gy; nz = g.gz;
    } else if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const int m0 = by * M2_BM;
    const int p0 = bx * WBN;
    const int ri = bz & 3, ks = bz >> 2, nsplit = nz >> 2;
    const int cpt = g.C / M2_KC;
    const int per_split = (cpt + nsplit - 1) / nsplit;
    const int ch_lo = ks * per_split;
    const int ch_hi = ch_lo + per_split < cpt ? ch_lo + per_split : cpt;
    const int nchunk = ch_hi > ch_lo ? ch_hi - ch_lo : 0;
    const bool refl = g.pad_mode == 1;
    // ---- weight loader: float4 a4 (of the chunk's 8 channels) of row ar, for each horizontal component
    const int a4 = tid & 1, ar = tid >> 1;
    int mrow = m0 + ar;
    mrow = mrow < g.M ? mrow : g.M - 1;
    const unsigned u_comp = 4u * (unsigned)g.M * 4u * (unsigned)g.C;
    const unsigned u_base = 4u * (((unsigned)mrow * 4u + (unsigned)ri) * (unsigned)g.C + (unsigned)(ch_lo * M2_KC) + 4u * a4);
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U);
    const __amdgpu_buffer_rsrc_t rsXd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    // ---- activation DMAs: piece L = 64 (wave + 4 q) + lane of the linear raw stream (8 rows x 34 pieces: pixels -4 .. 131 of the tile's
    //      flat pixel range over (image, tile row, x)); the two input rows of row component ri are fixed for the whole workgroup
    const int xr[2] = {ri == 0 ? 0 : (ri == 2 ? 2 : 1), ri == 3 ? 3 : (ri == 2 ? 1 : 2)};
    const int H2m2 = 2 * g.H - 2;
    unsigned d_row[2][2];                                        // [row set][q]: byte offset of the piece at channel 0 of the chunk, or FD_OOB
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int L = 64 * (wave + 4 * q) + lane;
        const int row = L / 34, seg = L - row * 34;
        const int F = 2 * p0 - 4 + 4 * seg;
        const bool ok = row < M2_KC && F >= 0 && F < g.Nb * HT * g.W;
        const int Fc = ok ? F : 0;
        const int nrow = Fc / g.W, x = Fc - nrow * g.W;
        const int n = nrow / HT, ty = nrow - n * HT;
        const unsigned base = 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)row * hw + (unsigned)x);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const int r = 2 * ty - 1 + xr[s_];
            const bool inb = (unsigned)r < (unsigned)g.H;
            int rr_ = r < 0 ? -r : r;
            rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
            const int ruse = refl ? rr_ : r;
            d_row[s_][q] = (ok & (refl | inb)) ? base + (unsigned)(ruse * g.W * 4) : FD_OOB;
        }
    }
    unsigned d_off[2][2] = {{FD_OOB, FD_OOB}, {FD_OOB, FD_OOB}};
    unsigned d_soff = 0u, u_off = FD_OOB;
    int pc = 0;                                                  // chunk (relative to ch_lo) the offsets point at
    auto prep = [&]() __attribute__((always_inline)) {           // offsets of chunk pc, then advance
        const bool live = pc < nchunk;
        u_off = live ? u_base + 4u * (unsigned)(pc * M2_KC) : FD_OOB;
        d_soff = 4u * (unsigned)((ch_lo + pc) * M2_KC) * hw;     // wave-uniform: first channel of the chunk
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int q = 0; q < 2; ++q) d_off[s_][q] = live ? d_row[s_][q] : FD_OOB;
        ++pc;
    };
    float4 ru[4];
    auto load_u = [&](int t) __attribute__((always_inline)) { ru[t] = fd_ldg128(rsU, u_off + (unsigned)t * u_comp); };
    auto store_u = [&](int buf, int t) __attribute__((always_inline)) {
        float* q = smem + buf * M2_BUF_FLOATS + t * M2_KC * M2_LDU + (4 * a4) * M2_LDU + ar;
        q[0] = ru[t].x; q[M2_LDU] = ru[t].y; q[2 * M2_LDU] = ru[t].z; q[3 * M2_LDU] = ru[t].w;
    };
    auto dma_v = [&](int buf, int s_, int q) __attribute__((always_inline)) {
        float* dst = smem + buf * M2_BUF_FLOATS + 4 * M2_KC * M2_LDU + s_ * M2_VRAW + (wave + 4 * q) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXd, (__attribute__((address_space(3))) void*)dst, 16, (int)d_off[s_][q], (int)d_soff, 0, 0);
    };

    const int wm = wave >> 1, wn = wave & 1;
    int o12, o0, o3;
    float ml, mr;
    {
        const int jp = 32 * wn + (lane & 31);
        const int pp = p0 + jp < Np ? p0 + jp : 0;
        const int rem = pp % plane2;
        const int jj = rem % W2;
        const bool le = jj == 0, re = 2 * jj + 2 >= g.W;
        o12 = 4 + 2 * jp;
        o0 = (le && refl) ? o12 : o12 - 2;       // 8-byte cell whose .y is d0 (reflection: column -1 is column 1 = d12.y)
        o3 = (re && refl) ? o12 : o12 + 2;       // 8-byte cell whose .x is d3 (reflection: column W is column W - 2 = d12.x)
        ml = (le && !refl) ? 0.f : 1.f;
        mr = (re && !refl) ? 0.f : 1.f;
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][t][r] = 0.f;

    constexpr int NK = M2_KC / 2;                                // 4 k-steps of 8 matrix instructions per chunk
    const int arow = lane >> 5, acol = lane & 31;
    float sgn = ri == 1 ? 1.f : -1.f;                            // the row combination: rowA + sgn * rowB
    asm volatile("" : "+v"(sgn));                                // in a VGPR: an SGPR operand halves the VALU rate on gfx950
    if (nchunk > 0) {
        prep();                                                  // chunk 0
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            dma_v(0, s_, 0);
            if (wave == 0) dma_v(0, s_, 1);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) store_u(0, t);
        prep();                                                  // offsets of chunk 1: fetched DURING chunk 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            const float* pa = smem + cur * M2_BUF_FLOATS + arow * M2_LDU + 64 * wm + acol;
            const float* pr = smem + cur * M2_BUF_FLOATS + 4 * M2_KC * M2_LDU + arow * LDR;
            typedef const __attribute__((address_space(3))) float* lds_cf;       // (stays an LDS pointer through the asm: a generic one
            typedef const __attribute__((address_space(3))) f32x2* lds_cf2;      //  turns the reads into flat loads)
            lds_cf pe = (lds_cf)(pr + M2_VRAW);                      // row set B through its own address register: with one base hipcc
            asm volatile("" : "+v"(pe));                             // pairs the reads into ds_read2st64_b64 (8 LDS cycles instead of 2 x 2)
            float av[2][2][4], bv[2][4];
            auto read_a = [&](int nb, int k2, int b, int t) __attribute__((always_inline)) { av[nb][b][t] = pa[t * M2_KC * M2_LDU + k2 * M2_LDU + 32 * b]; };
            f32x2 d12, e12, dl, dr, el, er;
            auto read_b = [&](int k2) __attribute__((always_inline)) {
                d12 = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o12);
                e12 = *(lds_cf2)(pe + k2 * LDR + o12);
                dl = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o0); dr = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o3);
                el = *(lds_cf2)(pe + k2 * LDR + o0); er = *(lds_cf2)(pe + k2 * LDR + o3);
            };
            auto xform_b = [&](int nb) __attribute__((always_inline)) {
                asm volatile("" : "+v"(dl), "+v"(dr), "+v"(el), "+v"(er));   // both halves live: keeps the reads 8 bytes wide
                const float c0 = fmaf(sgn, el.y, dl.y), c1 = fmaf(sgn, e12.x, d12.x), c2 = fmaf(sgn, e12.y, d12.y), c3 = fmaf(sgn, er.x, dr.x);
                bv[nb][0] = fmaf(c0, ml, -c2); bv[nb][1] = c1 + c2; bv[nb][2] = c2 - c1; bv[nb][3] = fmaf(-c3, mr, c1);
            };
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < 4; ++t) read_a(0, 0, b, t);
            read_b(0); xform_b(0);
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                const bool more = kk + 1 < NK;
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0][0], bv[cb][0], acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) read_b(2 * (kk + 1));
                __builtin_amdgcn_sched_barrier(0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0][1], bv[cb][1], acc[0][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { read_a(nb, 2 * (kk + 1), 0, 0); read_a(nb, 2 * (kk + 1), 0, 1); read_a(nb, 2 * (kk + 1), 0, 2); }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0][2], bv[cb][2], acc[0][2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { read_a(nb, 2 * (kk + 1), 0, 3); read_a(nb, 2 * (kk + 1), 1, 0); read_a(nb, 2 * (kk + 1), 1, 1); }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0][3], bv[cb][3], acc[0][3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) { read_a(nb, 2 * (kk + 1), 1, 2); read_a(nb, 2 * (kk + 1), 1, 3); }
                if (kk < 2) load_u(2 * kk);
                if (kk >= 2) store_u(cur ^ 1, 2 * (kk - 2));
                __builtin_amdgcn_sched_barrier(0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1][0], bv[cb][0], acc[1][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < 2) load_u(2 * kk + 1);
                if (kk >= 2) store_u(cur ^ 1, 2 * (kk - 2) + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1][1], bv[cb][1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (more) xform_b(nb);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1][2], bv[cb][2], acc[1][2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < 2) dma_v(cur ^ 1, kk, 0);                       // row set A, then row set B
                if (kk == 2 && wave == 0) { dma_v(cur ^ 1, 0, 1); dma_v(cur ^ 1, 1, 1); }
                if (kk == NK - 1) prep();                                // chunk ch + 2; every fetch of chunk ch + 1 has been issued by now
                __builtin_amdgcn_sched_barrier(0);
                acc[1][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1][3], bv[cb][3], acc[1][3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // this chunk's DMAs (into the other buffer) have landed
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    }

    // ---- epilogue: the horizontally transformed products S_ri [N][M][H/2][W] of this (row component, channel split) to slab bz
    const int po = p0 + 32 * wn + acol;
    const unsigned hwo = (unsigned)(HT * g.W);
    unsigned out_base = FD_OOB;
    if (po < Np) {
        const int n = po / plane2;
        const int rem = po - n * plane2;
        const int yy = rem / W2, jj = rem - yy * W2;
        out_base = 4u * ((unsigned)n * (unsigned)g.M * hwo + (unsigned)(yy * g.W + 2 * jj));
    }
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.slabs + (size_t)bz * g.slab_stride);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int mbase = m0 + 64 * wm + 32 * b + 4 * arow;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            const unsigned off = (m < g.M) ? out_base + 4u * (unsigned)m * hwo : FD_OOB;      // out of range: the store is dropped
            f32x2 o;
            o.x = (acc[b][0][r] + acc[b][1][r]) + acc[b][2][r];
            o.y = (acc[b][1][r] - acc[b][2][r]) - acc[b][3][r];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsY, (int)off, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------ F(2x2, 3x3) slabs, split precision
// k_conv_wino2d_limb: the slab kernel of the deep layers (k_conv_wino2d / _m128: matrix pipes 0.7 busy - matrix-bound, unlike the
// one-workgroup kernel, whose split-precision form gained nothing: profiles/round6_wino2p_limb.log) with a bf16x3 matrix loop at fp32
// accuracy (conv_limb.h).  Same grid (pixel tile, 64-channel tile, row component x channel split; XCD-aware), same raw activation DMAs
// (row component fixed per workgroup), same slabs and finish kernels.  Different:
//   * the weights arrive PRE-SPLIT (re-layout modes 11 / 12: wino_limb_piece) - a chunk's (16 input channels) 24 planes of 64 fragments
//     are plain 16-byte copies global -> registers -> LDS;
//   * the activations are combined, transformed and split ONCE per workgroup by a transform stage between two barriers - thread = (2x2
//     tile, four channels of the chunk): 24 eight-byte raw reads, 16 fused multiply-adds + 16 transform operations, 8 split2, 12 eight-byte
//     fragment stores;
//   * the matrix phase of a chunk is 24 fragment reads + 24 v_mfma_f32_32x32x16_bf16 per wave (768 matrix-pipe cycles; the f32 kernels
//     spend 2 048 on the same 16 channels x 32 x 32 x 4 components).
// One raw buffer (consumed before the first barrier, refilled by DMA during the matrix phase), one fragment buffer per operand: 68 KB.
typedef __bf16 wl_bf16x8 __attribute__((ext_vector_type(8)));
#define FD_WLIMB_MFMA6(ACC, AF, BF)                                                                                                   \
    do {                                                                                                                              \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[2]), __builtin_bit_cast(wl_bf16x8, BF[0]), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[0]), __builtin_bit_cast(wl_bf16x8, BF[2]), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[1]), __builtin_bit_cast(wl_bf16x8, BF[1]), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[1]), __builtin_bit_cast(wl_bf16x8, BF[0]), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[0]), __builtin_bit_cast(wl_bf16x8, BF[1]), ACC, 0, 0, 0); \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wl_bf16x8, AF[0]), __builtin_bit_cast(wl_bf16x8, BF[0]), ACC, 0, 0, 0); \
    } while (0)
constexpr int W2L_HPL = 64 * 16 + 64;                 // one K half of a (component, limb) plane: 64 rows / tiles x 16 B, padded
constexpr int W2L_PLANE = 2 * W2L_HPL;
constexpr int W2L_OP = 12 * W2L_PLANE;                // one operand: 4 components x 3 limbs
constexpr int W2L_RAW_BYTES = 2 * V_RAW_FLOATS * 4;   // two raw row sets of 16 channels
constexpr int W2L_LDS_BYTES = 2 * W2L_OP + W2L_RAW_BYTES;

__global__ void __launch_bounds__(WNT) k_conv_wino2d_limb(WinoArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* const smA = reinterpret_cast<unsigned char*>(smem);
    unsigned char* const smB = smA + W2L_OP;
    float* const raw = reinterpret_cast<float*>(smA + 2 * W2L_OP);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1, HT = g.H >> 1;
    const int plane2 = HT * W2;
    const int Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z, nz = gridDim.z;
    if (g.xcd_swizzle == 2) {                                    // all pixel tiles of a (channel tile, row component, split) slice on one XCD
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        bx = k % g.gx;
        const int sl = (k / g.gx) * 8 + xcd;
        by = sl % g.gy; bz = sl / g.gy; nz = g.gz;
    } else if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    const int m0 = by * WBM;
    const int p0 = bx * WBN;
    const int ri = bz & 3, ks = bz >> 2, nsplit = nz >> 2;
    const int cpt = g.C / WBKC;
    const int per_split = (cpt + nsplit - 1) / nsplit;
    const int ch_lo = ks * per_split;
    const int ch_hi = ch_lo + per_split < cpt ? ch_lo + per_split : cpt;
    const int nchunk = ch_hi > ch_lo ? ch_hi - ch_lo : 0;
    const bool refl = g.pad_mode == 1;
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U);
    const __amdgpu_buffer_rsrc_t rsXd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    // ---- raw activation DMAs: piece L = 64 (wave + 4 q) + lane of the linear raw stream (16 rows x 34 pieces: pixels -4 .. 131 of the tile's
    //      flat pixel range over (image, tile row, x)); the two input rows of row component ri are fixed for the whole workgroup
    const int xr[2] = {ri == 0 ? 0 : (ri == 2 ? 2 : 1), ri == 3 ? 3 : (ri == 2 ? 1 : 2)};
    const int H2m2 = 2 * g.H - 2;
    unsigned d_row[2][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int L = 64 * (wave + 4 * q) + lane;
        const int row = L / 34, seg = L - row * 34;
        const int F = 2 * p0 - 4 + 4 * seg;
        const bool ok = row < WBKC && F >= 0 && F < g.Nb * HT * g.W;
        const int Fc = ok ? F : 0;
        const int nrow = Fc / g.W, x = Fc - nrow * g.W;
        const int n = nrow / HT, ty = nrow - n * HT;
        const unsigned base = 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)row * hw + (unsigned)x);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            const int r = 2 * ty - 1 + xr[s_];
            const bool inb = (unsigned)r < (unsigned)g.H;
            int rr_ = r < 0 ? -r : r;
            rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
            const int ruse = refl ? rr_ : r;
            d_row[s_][q] = (ok & (refl | inb)) ? base + (unsigned)(ruse * g.W * 4) : FD_OOB;
        }
    }
    // ---- weight fragments of a chunk: 24 planes x 64 rows = 1 536 pieces, six per thread (piece tid + 256 i: plane (tid + 256 i) / 64)
    unsigned u_lane[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = tid + 256 * i;
        const int pl = q >> 6, row = q & 63;
        int m = m0 + row; m = m < g.M ? m : g.M - 1;
        u_lane[i] = 16u * ((unsigned)pl * (unsigned)g.M + (unsigned)m);
    }
    const unsigned u_chunk = 16u * 24u * (unsigned)g.M;               // bytes per chunk of the image
    int pc = 0;                                                       // chunk (relative to ch_lo) the next fetch takes
    uint4 ru[6];
    auto fetch = [&]() __attribute__((always_inline)) {               // weights of chunk pc -> registers, raw rows of chunk pc -> LDS; then advance
        const bool live = pc < nchunk;
        const unsigned u_soff = (unsigned)(ri * cpt + ch_lo + (live ? pc : 0)) * u_chunk;
        const unsigned d_soff = 4u * (unsigned)((ch_lo + pc) * WBKC) * hw;
#pragma unroll
        for (int i = 0; i < 6; ++i)
            ru[i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsU, (int)(live ? u_lane[i] : FD_OOB), (int)u_soff, 0));
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q == 2 && wave != 0) continue;
                float* dst = raw + s_ * V_RAW_FLOATS + (wave + 4 * q) * 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXd, (__attribute__((address_space(3))) void*)dst, 16, (int)(live ? d_row[s_][q] : FD_OOB), (int)d_soff, 0, 0);
            }
        ++pc;
    };
    auto store_u = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = tid + 256 * i;
            const int pl = q >> 6, row = q & 63;                      // plane = (t * 3 + L) * 2 + h
            *reinterpret_cast<uint4*>(smA + (pl >> 1) * W2L_PLANE + (pl & 1) * W2L_HPL + row * 16) = ru[i];
        }
    };
    // ---- transform stage: this thread's 2x2 tile (lane) and channels 4 wave .. 4 wave + 3 of the chunk
    int t12, t0, t3;
    float tml, tmr;
    {
        const int pp = p0 + lane < Np ? p0 + lane : 0;
        const int rem = pp % plane2;
        const int jj = rem % W2;
        const bool le = jj == 0, re = 2 * jj + 2 >= g.W;
        t12 = 4 + 2 * lane;
        t0 = (le && refl) ? t12 : t12 - 2;
        t3 = (re && refl) ? t12 : t12 + 2;
        tml = (le && !refl) ? 0.f : 1.f;
        tmr = (re && !refl) ? 0.f : 1.f;
    }
    float sgn = ri == 1 ? 1.f : -1.f;                                 // the row combination: rowA + sgn * rowB
    asm volatile("" : "+v"(sgn));
    unsigned char* const bslot = smB + (wave >> 1) * W2L_HPL + lane * 16 + 8 * (wave & 1);
    auto transform = [&]() __attribute__((always_inline)) {
        float v[4][4];                                               // [component][channel]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* pr = raw + (4 * wave + i) * LDR;
            const float* pe = pr + V_RAW_FLOATS;
            const f32x2 d12 = *reinterpret_cast<const f32x2*>(pr + t12), e12 = *reinterpret_cast<const f32x2*>(pe + t12);
            const f32x2 dl = *reinterpret_cast<const f32x2*>(pr + t0), dr = *reinterpret_cast<const f32x2*>(pr + t3);
            const f32x2 el = *reinterpret_cast<const f32x2*>(pe + t0), er = *reinterpret_cast<const f32x2*>(pe + t3);
            const float c0 = fmaf(sgn, el.y, dl.y), c1 = fmaf(sgn, e12.x, d12.x), c2 = fmaf(sgn, e12.y, d12.y), c3 = fmaf(sgn, er.x, dr.x);
            v[0][i] = fmaf(c0, tml, -c2); v[1][i] = c1 + c2; v[2][i] = c2 - c1; v[3][i] = fmaf(-c3, tmr, c1);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned h0, m0_, l0, h1, m1, l1;
            fdlimb::split2(v[t][0], v[t][1], h0, m0_, l0); fdlimb::split2(v[t][2], v[t][3], h1, m1, l1);
            *reinterpret_cast<u32x2*>(bslot + (3 * t) * W2L_PLANE) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(bslot + (3 * t + 1) * W2L_PLANE) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(bslot + (3 * t + 2) * W2L_PLANE) = u32x2{l0, l1};
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int arow = lane >> 5, acol = lane & 31;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    if (nchunk > 0) {
        fetch();                                                     // chunk 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const unsigned char* fa = smA + arow * W2L_HPL + (32 * wm + acol) * 16;
        const unsigned char* fb = smB + arow * W2L_HPL + (32 * wn + acol) * 16;
        for (int ch = 0; ch < nchunk; ++ch) {
            store_u();                                               // weight fragments of chunk ch (in registers since the last matrix phase)
            transform();                                             // raw rows of chunk ch -> activation fragments
            __syncthreads();                                         // fragments complete, raw buffer free
            fetch();                                                 // chunk ch + 1 (past the end: nothing is fetched)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint4 af[3], bf[3];
#pragma unroll
                for (int Lm = 0; Lm < 3; ++Lm) {
                    af[Lm] = *reinterpret_cast<const uint4*>(fa + (t * 3 + Lm) * W2L_PLANE);
                    bf[Lm] = *reinterpret_cast<const uint4*>(fb + (t * 3 + Lm) * W2L_PLANE);
                }
                FD_WLIMB_MFMA6(acc[t], af, bf);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the next chunk's DMAs and weight loads have landed
            __syncthreads();
        }
    }
    // ---- epilogue (k_conv_wino2d_m128's, one 32-row block per wave): S_ri [N][M][H/2][W] of this (row component, channel split) to slab bz
    const int po = p0 + 32 * wn + acol;
    const unsigned hwo = (unsigned)(HT * g.W);
    unsigned out_base = FD_OOB;
    if (po < Np) {
        const int n = po / plane2;
        const int rem = po - n * plane2;
        const int yy = rem / W2, jj = rem - yy * W2;
        out_base = 4u * ((unsigned)n * (unsigned)g.M * hwo + (unsigned)(yy * g.W + 2 * jj));
    }
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.slabs + (size_t)bz * g.slab_stride);
    const int mbase = m0 + 32 * wm + 4 * arow;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2);
        const unsigned off = (m < g.M) ? out_base + 4u * (unsigned)m * hwo : FD_OOB;
        f32x2 o;
        o.x = (acc[0][r] + acc[1][r]) + acc[2][r];
        o.y = (acc[1][r] - acc[2][r]) - acc[3][r];
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsY, (int)off, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------ F(2x2, 3x3) in ONE workgroup
// k_conv_wino2p (round 4): the 16 components of F(2x2, 3x3) for the layers where slabs cost more than the matrix work they save
// (ResNet layer1 / layer2, the decoder's wide blocks: few channels, many pixels - the 1-D kernel's territory until now).  A
// workgroup keeps its 64 (channels) x 64 (2x2 tiles) output tile for the WHOLE computation and runs the four row components one
// after the other through the same four horizontal accumulators: GEMM-K = (ri, input channel), a flat sequence of 4 C / 16 chunks
// fed by one uninterrupted register-staged pipeline (the 2-D loader of k_conv_wino2d with ri advancing per chunk).  At the end of
// a row component the horizontal output transform h = (M0 + M1 + M2, M1 - M2 - M3) is folded into two more accumulator pairs - the
// vertical output transform y[2 ty] = S0 + S1 + S2, y[2 ty + 1] = S1 - S2 - S3 is linear - and the epilogue writes FINAL outputs
// (bias, activation, residual, BatchNorm partial sums): no slabs, no finishing launch, 16 instead of 24 products per 2x2 tile,
// K loops a third longer than the 1-D kernel's (4 C / 16 against 3 C / 16 chunks per tile of twice the pixels), and the four row
// combinations of a tile's input rows are formed by ONE workgroup out of L1 / L2 instead of four.  128 accumulator registers per
// lane: two waves per SIMD.
#ifndef FD_W2P_ABLATE
#define FD_W2P_ABLATE 0      // timing experiments only (wrong results): 1 no fold, 2 loop loads out of range, 4 no LDS stores in the loop, 8 no barrier, 16 no output stores
#endif
typedef float f32x16_w2p __attribute__((ext_vector_type(16)));
// Final outputs of a k_conv_wino2p tile from the two folded accumulator pairs (shared by the register-staged and the direct-to-LDS
// variant of the kernel)
template <bool STATS>
__device__ __forceinline__ void w2p_epilogue(const WinoArgs& g, const f32x16_w2p (&ya)[2], const f32x16_w2p (&yb)[2], int p0, int m0, int Np, int plane2,
                                             int W2, unsigned hw, int lane, int wm, int wn) {
    const int arow = lane >> 5, acol = lane & 31;
    // ---- epilogue: final outputs of the tile's two rows (C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5))
    const int po = p0 + 32 * wn + acol;
    unsigned out_base = FD_OOB;
    if (po < Np) {
        const int n = po / plane2;
        const int rem = po - n * plane2;
        const int yy = rem / W2, jj = rem - yy * W2;
        out_base = 4u * ((unsigned)n * (unsigned)g.M * hw + (unsigned)(2 * yy * g.W + 2 * jj));
    }
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.Y);
    const __amdgpu_buffer_rsrc_t rsAdd = fd_make_rsrc(g.add ? g.add : g.Y);
    const bool has_add = g.add != nullptr;
    const int mbase = m0 + 32 * wm + 4 * arow;
    const unsigned row_b = 4u * (unsigned)g.W;
    float bias_r[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bias_r[r] = 0.f;
    if (g.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            bias_r[r] = g.bias[m < g.M ? m : g.M - 1];
        }
    }
    float s1[16], s2[16];                       // STATS: (sum, M2) of each row's four pixels
    auto rows = [&](auto act_tag) __attribute__((always_inline)) {
        constexpr int ACT = decltype(act_tag)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2);
            const unsigned off = (m < g.M) ? out_base + 4u * (unsigned)m * hw : FD_OOB;
            f32x2 oa, ob;
            oa.x = ya[0][r] + bias_r[r]; oa.y = ya[1][r] + bias_r[r];
            ob.x = yb[0][r] + bias_r[r]; ob.y = yb[1][r] + bias_r[r];
            if (ACT != 0) { oa.x = wino_act(oa.x, g.act); oa.y = wino_act(oa.y, g.act); ob.x = wino_act(ob.x, g.act); ob.y = wino_act(ob.y, g.act); }
            if (has_add) {
                const f32x2 a2 = fd_ldg64(rsAdd, off), b2 = fd_ldg64(rsAdd, off + row_b);
                oa.x += a2.x; oa.y += a2.y; ob.x += b2.x; ob.y += b2.y;
            }
            if (!(FD_W2P_ABLATE & 16) || oa.x == 123.456f) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oa), rsY, (int)off, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rsY, (int)(off + row_b), 0, 0);
            }
            if (STATS) {
                const float da = oa.x - oa.y, db = ob.x - ob.y;
                const float ta = oa.x + oa.y, tb = ob.x + ob.y, dab = ta - tb;
                s1[r] = ta + tb;
                s2[r] = fmaf(dab * dab, 0.25f, 0.5f * da * da + 0.5f * db * db);      // two pairs of two: (sa - sb)^2 / (2 * 2)
            }
        }
    };
    if (g.act == 0) rows(std::integral_constant<int, 0>{});
    else rows(std::integral_constant<int, -1>{});
    if (STATS) {
        // the butterfly of k_conv_wino with four pixels per lane and row to start from: slots of 32 tiles = 128 pixels
#pragma unroll
        for (int step = 0; step < 4; ++step) {
            const int width = 8 >> step;
            const int xm = 16 >> step;
            const bool hi = (lane & xm) != 0;
            const float inv2n = 0.125f / (float)(1 << step);                // each side holds n = 4 << step pixels
#pragma unroll
            for (int j = 0; j < width; ++j) {
                const float k1 = hi ? s1[j + width] : s1[j], g1 = hi ? s1[j] : s1[j + width];
                const float k2 = hi ? s2[j + width] : s2[j], g2 = hi ? s2[j] : s2[j + width];
                const float o1 = __shfl_xor(g1, xm, 64), o2 = __shfl_xor(g2, xm, 64);
                const float df = k1 - o1;
                s1[j] = k1 + o1;
                s2[j] = fmaf(df * df, inv2n, k2 + o2);
            }
        }
        {
            const float o1 = __shfl_xor(s1[0], 1, 64), o2 = __shfl_xor(s2[0], 1, 64);
            const float df = s1[0] - o1;
            s2[0] = fmaf(df * df, 1.0f / 128.0f, s2[0] + o2);               // n = 64 per side
            s1[0] += o1;
        }
        const int rr = ((lane >> 4) & 1) * 8 + ((lane >> 3) & 1) * 4 + ((lane >> 2) & 1) * 2 + ((lane >> 1) & 1);
        const int m = mbase + (rr & 3) + 8 * (rr >> 2);
        const int n = p0 / plane2, tile = (p0 - n * plane2) / WBN;         // the whole tile lies in image n (launcher's guarantee)
        if (!(lane & 1) && m < g.M && 2 * tile + wn < g.stat_slots) {      // (image-aligned tiling: the last tile's second half may be empty)
            f32x2 v; v.x = s1[0]; v.y = s2[0];
            *reinterpret_cast<f32x2*>(g.stat_part + (((size_t)n * g.M + m) * g.stat_slots + 2 * tile + wn) * 2) = v;
        }
    }
}

template <bool STATS>
__global__ void __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_conv_wino2p(WinoArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1, HT = g.H >> 1;
    const int plane2 = HT * W2;                                          // 2x2 tiles per image; Nb * plane2 < 2^29 (size guard)
    const int Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    int bx = blockIdx.x;
    const int by = blockIdx.y;
    if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }     // vertical neighbours share input rows: same XCD
    const int m0 = by * WBM, p0 = bx * WBN;
    const int cpt = g.C / WBKC, nchunk = 4 * cpt;
    auto ri_of = [&](int f) __attribute__((always_inline)) { return (f >= cpt ? 1 : 0) + (f >= 2 * cpt ? 1 : 0) + (f >= 3 * cpt ? 1 : 0); };

    // ---- activation loader: this thread always fetches tile jn of the workgroup's 64, channel rows kr + 4 i
    const int jn = lane, kr = wave;
    const int pg = p0 + jn;
    const bool pvalid = pg < Np;
    int y0, j0;
    unsigned nbase;
    {
        const int pp = pvalid ? pg : 0;
        const int n = pp / plane2;
        const int rem = pp - n * plane2;
        y0 = rem / W2; j0 = rem - y0 * W2;
        nbase = (unsigned)n * (unsigned)g.C * hw;
    }
    const bool refl = g.pad_mode == 1;
    const bool left_edge = j0 == 0, right_edge = 2 * j0 + 2 >= g.W;
    const bool halo_l = jn == 0 && !left_edge, halo_r = jn == WBN - 1 && !right_edge;
    const int a4 = tid & 3, ar = tid >> 2;
    int mrow = m0 + ar;
    mrow = mrow < g.M ? mrow : g.M - 1;
    const unsigned u_comp = 4u * (unsigned)g.M * 4u * (unsigned)g.C;     // bytes between components of U2[t][m][ri][c]
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U), rsX = fd_make_rsrc(g.X);
    float4 ru[4];
    f32x2 rmid[4], rmid2[4];
    float rh[4], rh2[4];
    unsigned u_off = FD_OOB, mid_off = FD_OOB, h_off = FD_OOB, mid_off2 = FD_OOB, h_off2 = FD_OOB;
    const unsigned c_step = 4u * 4u * hw;
    int pc_f = 0;                                                        // flat chunk index being prepared
    int pc_ri = 0, pc_c0 = 0;
    unsigned prep_base = 0u, prep_base2 = 0u;
    bool prep_ok = false, prep_ok2 = false;
    const int H2m2 = 2 * g.H - 2;
    auto prep_a = [&]() __attribute__((always_inline)) {
        const bool live = pc_f < nchunk;
        const int ri = pc_ri;
        const int xr_a = ri == 0 ? 0 : (ri == 2 ? 2 : 1), xr_b = ri == 3 ? 3 : (ri == 2 ? 1 : 2);
        u_off = live ? 4u * (((unsigned)mrow * 4u + (unsigned)ri) * (unsigned)g.C + (unsigned)pc_c0 + 4u * a4) : FD_OOB;
        auto row = [&](int r, bool& ok, unsigned& base) __attribute__((always_inline)) {
            const bool inb = (unsigned)r < (unsigned)g.H;
            int rr_ = r < 0 ? -r : r;
            rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
            const int ruse = refl ? rr_ : r;
            ok = pvalid & live & (refl | inb);
            base = 4u * (nbase + (unsigned)(pc_c0 + kr) * hw + (unsigned)(ruse * g.W + 2 * j0));
        };
        row(2 * y0 - 1 + xr_a, prep_ok, prep_base);
        row(2 * y0 - 1 + xr_b, prep_ok2, prep_base2);
    };
    auto prep_b = [&]() __attribute__((always_inline)) {
        mid_off = prep_ok ? prep_base : FD_OOB;
        h_off = (prep_ok & halo_l) ? prep_base - 4u : ((prep_ok & halo_r) ? prep_base + 8u : FD_OOB);
        mid_off2 = prep_ok2 ? prep_base2 : FD_OOB;
        h_off2 = (prep_ok2 & halo_l) ? prep_base2 - 4u : ((prep_ok2 & halo_r) ? prep_base2 + 8u : FD_OOB);
        if ((FD_W2P_ABLATE & 2) && pc_f > 1) { u_off = mid_off = h_off = mid_off2 = h_off2 = FD_OOB; }
        ++pc_f;
        pc_c0 += WBKC;
        const bool wrap = pc_c0 >= g.C;
        pc_c0 = wrap ? 0 : pc_c0;
        pc_ri += wrap ? 1 : 0;
    };
    auto load_u = [&](int t) __attribute__((always_inline)) { ru[t] = fd_ldg128(rsU, u_off + (unsigned)t * u_comp); };
    auto load_mid = [&](int i) __attribute__((always_inline)) {
        rmid[i] = fd_ldg64(rsX, mid_off + (unsigned)i * c_step);
        rmid2[i] = fd_ldg64(rsX, mid_off2 + (unsigned)i * c_step);
    };
    auto load_h = [&](int i) __attribute__((always_inline)) {
        rh[i] = fd_ldg32(rsX, h_off + (unsigned)i * c_step);
        rh2[i] = fd_ldg32(rsX, h_off2 + (unsigned)i * c_step);
    };
    auto store_u = [&](int buf, int t) __attribute__((always_inline)) {
        float* q = smem + buf * W_BUF_FLOATS + t * WBKC * LDU + (4 * a4) * LDU + ar;
        q[0] = ru[t].x; q[LDU] = ru[t].y; q[2 * LDU] = ru[t].z; q[3 * LDU] = ru[t].w;
    };
    const int v_row = 4 * WBKC * LDU + kr * LDR;
    const int h_col = jn == 0 ? 3 : 2 * WBN + 4;
    // `sgn`: sign of the second row in the row combination of the chunk whose data sit in the staging registers (+1 for ri = 1)
    auto store_v = [&](int buf, int i, float sgn) __attribute__((always_inline)) {
        float* q = smem + buf * W_BUF_FLOATS + v_row + 4 * i * LDR;
        rmid[i].x = fmaf(sgn, rmid2[i].x, rmid[i].x); rmid[i].y = fmaf(sgn, rmid2[i].y, rmid[i].y);
        rh[i] = fmaf(sgn, rh2[i], rh[i]);
        *reinterpret_cast<f32x2*>(q + 4 + 2 * jn) = rmid[i];
        if (jn == 0 || jn == WBN - 1) q[h_col] = rh[i];
    };

    const int wm = wave >> 1, wn = wave & 1;
    int o12, o0, o3;
    float ml, mr;
    {
        const int jp = 32 * wn + (lane & 31);
        const int pp = p0 + jp < Np ? p0 + jp : 0;
        const int rem = pp % plane2;
        const int jj = rem % W2;
        const bool le = jj == 0, re = 2 * jj + 2 >= g.W;
        o12 = 4 + 2 * jp;
        o0 = (le && refl) ? o12 : o12 - 2;       // 8-byte cell whose .y is d0 (reflection: column -1 is column 1 = d12.y)
        o3 = (re && refl) ? o12 : o12 + 2;       // 8-byte cell whose .x is d3 (reflection: column W is column W - 2 = d12.x)
        ml = (le && !refl) ? 0.f : 1.f;
        mr = (re && !refl) ? 0.f : 1.f;
    }
    f32x16 acc[4], ya[2], yb[2];                 // working components; output rows 2 ty (ya) / 2 ty + 1 (yb), columns 2 j / 2 j + 1
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][r] = 0.f;
        ya[0][r] = ya[1][r] = yb[0][r] = yb[1][r] = 0.f;
    }

    constexpr int NK = WBKC / 2, LS = NK / 2;
    const int arow = lane >> 5, acol = lane & 31;
    {
        prep_a(); prep_b();
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load_mid(i); load_h(i); }
#pragma unroll
        for (int t = 0; t < 4; ++t) store_u(0, t);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_v(0, i, -1.f);             // chunk 0 is ri = 0: x_r0 - x_r2
        prep_a(); prep_b();                                          // chunk 1: loaded now, written to LDS during chunk 0
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
        for (int i = 0; i < 4; ++i) { load_mid(i); load_h(i); }
        prep_a(); prep_b();                                          // offsets of chunk 2, re-loaded during chunk 0
        __syncthreads();
        int next_fold = cpt - 1;                                     // last chunk of the current row component
        int ri_cur = 0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            const float sgn_regs = ri_of(ch + 1) == 1 ? 1.f : -1.f;  // the registers hold chunk ch + 1
            const float* pa = smem + cur * W_BUF_FLOATS + arow * LDU + 32 * wm + acol;
            const float* pr = smem + cur * W_BUF_FLOATS + 4 * WBKC * LDU + arow * LDR;
            float av[2][4], bv[2][4];
            auto read_a = [&](int nb, int k2, int t) __attribute__((always_inline)) { av[nb][t] = pa[t * WBKC * LDU + k2 * LDU]; };
            f32x2 d12, dl, dr;                                           // three 8-byte reads (conflict-free at stride 8 over a half-wave;
            auto read_b = [&](int k2) __attribute__((always_inline)) {   // the 4-byte reads of d0 / d3 at stride 8 were 2-way bank conflicts)
                d12 = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o12);
                dl = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o0); dr = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o3);
            };
            auto xform_b = [&](int nb) __attribute__((always_inline)) {
                asm volatile("" : "+v"(dl), "+v"(dr));                   // both halves live: keeps the reads 8 bytes wide
                bv[nb][0] = fmaf(dl.y, ml, -d12.y); bv[nb][1] = d12.x + d12.y; bv[nb][2] = d12.y - d12.x; bv[nb][3] = fmaf(-dr.x, mr, d12.x);
            };
#pragma unroll
            for (int t = 0; t < 4; ++t) read_a(0, 0, t);
            read_b(0); xform_b(0);
#pragma unroll
            for (int kk = 0; kk < NK; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NK) { read_b(2 * (kk + 1)); read_a(nb, 2 * (kk + 1), 0); read_a(nb, 2 * (kk + 1), 1); }
                if (kk < LS && !(FD_W2P_ABLATE & 4)) store_u(cur ^ 1, kk);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NK) { read_a(nb, 2 * (kk + 1), 2); read_a(nb, 2 * (kk + 1), 3); }
                if (kk < LS) load_u(kk);
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][2], bv[cb][2], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < LS && !(FD_W2P_ABLATE & 4)) store_v(cur ^ 1, kk, sgn_regs);
                if (kk + 1 < NK) xform_b(nb);
                __builtin_amdgcn_sched_barrier(0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][3], bv[cb][3], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < LS) { load_mid(kk); load_h(kk); }
                if (kk == NK - 2) prep_a();                          // chunk ch + 3; every load of chunk ch + 2 has been issued by now
                if (kk == NK - 1) prep_b();
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ch == next_fold && !(FD_W2P_ABLATE & 1)) {
                // end of row component ri_cur: fold the horizontally transformed products into the two output rows and start afresh
                const float sa = ri_cur <= 2 ? 1.f : 0.f;             // y[2 ty]     = S0 + S1 + S2
                const float sb = ri_cur == 0 ? 0.f : (ri_cur == 1 ? 1.f : -1.f);   // y[2 ty + 1] = S1 - S2 - S3
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h0 = (acc[0][r] + acc[1][r]) + acc[2][r];
                    const float h1 = (acc[1][r] - acc[2][r]) - acc[3][r];
                    ya[0][r] = fmaf(sa, h0, ya[0][r]); ya[1][r] = fmaf(sa, h1, ya[1][r]);
                    yb[0][r] = fmaf(sb, h0, yb[0][r]); yb[1][r] = fmaf(sb, h1, yb[1][r]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][r] = 0.f;
                }
                next_fold += cpt;
                ++ri_cur;
            }
            if (!(FD_W2P_ABLATE & 8)) __syncthreads();
        }
    }
    w2p_epilogue<STATS>(g, ya, yb, p0, m0, Np, plane2, W2, hw, lane, wm, wn);
}

// The same computation with the activations on the LDS-DMA path (needs W % 4 == 0, a 16-byte aligned tensor): BOTH input rows of the
// chunk's row combination go from global memory straight into LDS as raw rows (`buffer_load_dwordx4 ... lds`, three 1-KB pieces per
// wave and row set), and the vertical AND the horizontal input transform are applied when the B operands are read:
//   d = rowA + sgn * rowB  (4 fused multiply-adds),  V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3).
// No staging registers, no s_waitcnt + ds_write for the activations in the MFMA stream (the VGPR -> LDS stores were the largest
// removable term of the register-staged loop: profiles/round4_w2p_ablation.md); 70 KB of LDS: two workgroups per CU, which is what
// the 128 accumulator registers allow anyway.  Flat pixel order of the DMA pieces: (image, tile row, x) - 64 consecutive 2x2 tiles are
// 128 consecutive columns of one tile row (or wrap into the next), exactly the 1-D kernel's scheme with H / 2 rows.
constexpr int W2D_BUF_FLOATS = 4 * WBKC * LDU + 2 * V_RAW_FLOATS;
constexpr int W2D_LDS_FLOATS = 2 * W2D_BUF_FLOATS;
// HALFM (round 5): at most 32 output channels (upconv(1, *) of the depth decoder) - the tile's rows 32 .. 63 do not exist, so the wave pair that
// would own them (wm = 1) takes the second half of every chunk's k-steps of rows 0 .. 31, and the two partial results (the folded accumulator
// pairs) meet in LDS once, in front of the epilogue, in a fixed order.
template <bool STATS, bool HALFM = false>
__global__ void __launch_bounds__(WNT) __attribute__((amdgpu_waves_per_eu(2, 2))) k_conv_wino2p_dma(WinoArgs g) {
    static_assert(!(STATS && HALFM), "no statistics epilogue for the 32-channel variant");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1, HT = g.H >> 1;
    const int plane2 = HT * W2;
    const int Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    int bx = blockIdx.x;
    const int by = blockIdx.y;
    if (g.xcd_swizzle) { const int per = gridDim.x >> 3; bx = (bx & 7) * per + (bx >> 3); }
    // g.img_tiles > 0: image-aligned tiling - workgroup bx is tile bx % img_tiles of image bx / img_tiles, so that no tile straddles
    // two images (the BatchNorm partial sums are per image) although the image's tile count is not a multiple of 64; tiles past the
    // image's last one (`lim`) are computed on whatever the loads return and never stored
    const int m0 = by * WBM;
    const int p0 = g.img_tiles > 0 ? (bx / g.img_tiles) * plane2 + (bx % g.img_tiles) * WBN : bx * WBN;
    const int lim = g.img_tiles > 0 ? (bx / g.img_tiles + 1) * plane2 : Np;
    const int cpt = g.C / WBKC, nchunk = 4 * cpt;
    const bool refl = g.pad_mode == 1;
    const int a4 = tid & 3, ar = tid >> 2;
    int mrow = m0 + ar;
    mrow = mrow < g.M ? mrow : g.M - 1;
    const unsigned u_comp = 4u * (unsigned)g.M * 4u * (unsigned)g.C;
    const __amdgpu_buffer_rsrc_t rsU = fd_make_rsrc(g.U);
    const __amdgpu_buffer_rsrc_t rsXd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    // per lane and DMA piece, fixed for the whole tile: tile row and byte offset (channel 0, image row 0) of its four pixels
    unsigned d_base[3] = {FD_OOB, FD_OOB, FD_OOB};
    int d_ty[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int L = 64 * (wave + 4 * q) + lane;
        const int row = L / 34, seg = L - row * 34;
        const int F = 2 * p0 - 4 + 4 * seg;                              // flat pixel index over (image, tile row, x)
        const bool ok = row < WBKC && F >= 0 && F < g.Nb * HT * g.W;
        const int Fc = ok ? F : 0;
        const int nrow = Fc / g.W, x = Fc - nrow * g.W;
        const int n = nrow / HT;
        d_ty[q] = nrow - n * HT;
        d_base[q] = ok ? 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)row * hw + (unsigned)x) : FD_OOB;
    }
    unsigned d_off[2][3] = {{FD_OOB, FD_OOB, FD_OOB}, {FD_OOB, FD_OOB, FD_OOB}};
    unsigned d_soff = 0u, u_off = FD_OOB;
    float4 ru[4];
    int pc_f = 0, pc_ri = 0, pc_c0 = 0;
    const int H2m2 = 2 * g.H - 2;
    auto prep = [&]() __attribute__((always_inline)) {                  // offsets of flat chunk pc_f, then advance
        const bool live = pc_f < nchunk;
        const int ri = pc_ri;
        const int xr[2] = {ri == 0 ? 0 : (ri == 2 ? 2 : 1), ri == 3 ? 3 : (ri == 2 ? 1 : 2)};
        u_off = live ? 4u * (((unsigned)mrow * 4u + (unsigned)ri) * (unsigned)g.C + (unsigned)pc_c0 + 4u * a4) : FD_OOB;
        d_soff = 4u * (unsigned)pc_c0 * hw;                              // wave-uniform: first channel of the chunk
        if (pc_c0 == 0) {                                                // the pieces' row offsets change with the row component only
#pragma unroll
            for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int r = 2 * d_ty[q] - 1 + xr[s_];
                    const bool inb = (unsigned)r < (unsigned)g.H;
                    int rr_ = r < 0 ? -r : r;
                    rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
                    const int ruse = refl ? rr_ : r;
                    d_off[s_][q] = (live & (refl | inb)) ? d_base[q] + (unsigned)(ruse * g.W * 4) : FD_OOB;   // FD_OOB base + anything stays out of range
                }
        }
        if ((FD_W2P_ABLATE & 2) && pc_f > 1) { u_off = FD_OOB; for (int s_ = 0; s_ < 2; ++s_) for (int q = 0; q < 3; ++q) d_off[s_][q] = FD_OOB; }
        ++pc_f;
        pc_c0 += WBKC;
        const bool wrap = pc_c0 >= g.C;
        pc_c0 = wrap ? 0 : pc_c0;
        pc_ri += wrap ? 1 : 0;
    };
    const bool u_rows = !HALFM || wave < 2;                          // HALFM: weight rows 0 .. 31 = the loader threads of waves 0 and 1
    auto load_u = [&](int t) __attribute__((always_inline)) { if (u_rows) ru[t] = fd_ldg128(rsU, u_off + (unsigned)t * u_comp); };
    auto store_u = [&](int buf, int t) __attribute__((always_inline)) {
        if (!u_rows) return;
        float* q = smem + buf * W2D_BUF_FLOATS + t * WBKC * LDU + (4 * a4) * LDU + ar;
        q[0] = ru[t].x; q[LDU] = ru[t].y; q[2 * LDU] = ru[t].z; q[3 * LDU] = ru[t].w;
    };
    auto dma_v = [&](int buf, int s_, int q) __attribute__((always_inline)) {
        float* dst = smem + buf * W2D_BUF_FLOATS + 4 * WBKC * LDU + s_ * V_RAW_FLOATS + (wave + 4 * q) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsXd, (__attribute__((address_space(3))) void*)dst, 16, (int)d_off[s_][q], (int)d_soff, 0, 0);
    };

    const int wm = wave >> 1, wn = wave & 1;
    int o12, o0, o3;
    float ml, mr;
    {
        const int jp = 32 * wn + (lane & 31);
        const int pp = p0 + jp < lim ? p0 + jp : 0;
        const int rem = pp % plane2;
        const int jj = rem % W2;
        const bool le = jj == 0, re = 2 * jj + 2 >= g.W;
        o12 = 4 + 2 * jp;
        o0 = (le && refl) ? o12 : o12 - 2;       // 8-byte cell whose .y is d0 (reflection: column -1 is column 1 = d12.y)
        o3 = (re && refl) ? o12 : o12 + 2;       // 8-byte cell whose .x is d3 (reflection: column W is column W - 2 = d12.x)
        ml = (le && !refl) ? 0.f : 1.f;
        mr = (re && !refl) ? 0.f : 1.f;
    }
    f32x16 acc[4], ya[2], yb[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t][r] = 0.f;
        ya[0][r] = ya[1][r] = yb[0][r] = yb[1][r] = 0.f;
    }

    constexpr int NK = WBKC / 2, LS = NK / 2;
    constexpr int NKW = HALFM ? NK / 2 : NK;                         // k-steps per wave and chunk (HALFM: wave pair wm takes k-steps NKW wm ...)
    const int kb2 = HALFM ? 2 * NKW * wm : 0;                        // first LDS operand row (= input channel of the chunk) of this wave's k-steps
    const int arow = lane >> 5, acol = lane & 31;
    {
        prep();                                                      // chunk 0
#pragma unroll
        for (int t = 0; t < 4; ++t) load_u(t);
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_) {
            dma_v(0, s_, 0); dma_v(0, s_, 1);
            if (wave == 0) dma_v(0, s_, 2);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) store_u(0, t);
        prep();                                                      // offsets of chunk 1: fetched DURING chunk 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int next_fold = cpt - 1, ri_cur = 0;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            float sgn = ri_cur == 1 ? 1.f : -1.f;                    // this chunk's row combination: rowA + sgn * rowB
            asm volatile("" : "+v"(sgn));                            // in a VGPR: an SGPR operand halves the VALU rate on gfx950
            const float* pa = smem + cur * W2D_BUF_FLOATS + (arow + kb2) * LDU + (HALFM ? 0 : 32 * wm) + acol;
            const float* pr = smem + cur * W2D_BUF_FLOATS + 4 * WBKC * LDU + (arow + kb2) * LDR;
            typedef const __attribute__((address_space(3))) float* lds_cf;       // (stays an LDS pointer through the asm: a generic one
            typedef const __attribute__((address_space(3))) f32x2* lds_cf2;      //  turns the reads into flat loads)
            lds_cf pe = (lds_cf)(pr + V_RAW_FLOATS);                     // row set B through its own address register: with one base hipcc
            asm volatile("" : "+v"(pe));                                 // pairs the reads into ds_read2st64_b64 (8 LDS cycles instead of 2 x 2)
            float av[2][4], bv[2][4];
            auto read_a = [&](int nb, int k2, int t) __attribute__((always_inline)) { av[nb][t] = pa[t * WBKC * LDU + k2 * LDU]; };
            f32x2 d12, e12, dl, dr, el, er;                              // 8-byte reads only (d0 / d3 as 4-byte reads at stride 8: 2-way bank conflicts)
            auto read_b = [&](int k2) __attribute__((always_inline)) {
                d12 = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o12);
                e12 = *(lds_cf2)(pe + k2 * LDR + o12);
                dl = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o0); dr = *reinterpret_cast<const f32x2*>(pr + k2 * LDR + o3);
                el = *(lds_cf2)(pe + k2 * LDR + o0); er = *(lds_cf2)(pe + k2 * LDR + o3);
            };
            auto xform_b = [&](int nb) __attribute__((always_inline)) {
                asm volatile("" : "+v"(dl), "+v"(dr), "+v"(el), "+v"(er));   // both halves live: keeps the reads 8 bytes wide
                const float c0 = fmaf(sgn, el.y, dl.y), c1 = fmaf(sgn, e12.x, d12.x), c2 = fmaf(sgn, e12.y, d12.y), c3 = fmaf(sgn, er.x, dr.x);
                bv[nb][0] = fmaf(c0, ml, -c2); bv[nb][1] = c1 + c2; bv[nb][2] = c2 - c1; bv[nb][3] = fmaf(-c3, mr, c1);
            };
#pragma unroll
            for (int t = 0; t < 4; ++t) read_a(0, 0, t);
            read_b(0); xform_b(0);
#pragma unroll
            for (int kk = 0; kk < NKW; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NKW) { read_b(2 * (kk + 1)); read_a(nb, 2 * (kk + 1), 0); read_a(nb, 2 * (kk + 1), 1); }
                if (!HALFM && kk >= LS && !(FD_W2P_ABLATE & 4)) store_u(cur ^ 1, kk - LS);
                if (HALFM && kk >= 2) { store_u(cur ^ 1, 2 * (kk - 2)); store_u(cur ^ 1, 2 * (kk - 2) + 1); }     // loaded in k-steps 0, 1
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NKW) { read_a(nb, 2 * (kk + 1), 2); read_a(nb, 2 * (kk + 1), 3); }
                if (!HALFM && kk < LS) load_u(kk);
                if (HALFM && kk < 2) { load_u(2 * kk); load_u(2 * kk + 1); }
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][2], bv[cb][2], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NKW) xform_b(nb);
                __builtin_amdgcn_sched_barrier(0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][3], bv[cb][3], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (!HALFM) {
                    if (kk < 4) dma_v(cur ^ 1, kk >> 1, kk & 1);         // row set A: pieces 0, 1; row set B: pieces 0, 1
                    if (kk == 4 && wave == 0) { dma_v(cur ^ 1, 0, 2); dma_v(cur ^ 1, 1, 2); }
                    if (kk == NK - 2) prep();                            // chunk ch + 2; every fetch of chunk ch + 1 has been issued by now
                } else {                                                 // the same six fetches in three k-steps, the offsets of chunk ch + 2 in the fourth
                    if (kk < 2) { dma_v(cur ^ 1, kk, 0); dma_v(cur ^ 1, kk, 1); }
                    if (kk == 2 && wave == 0) { dma_v(cur ^ 1, 0, 2); dma_v(cur ^ 1, 1, 2); }
                    if (kk == 3) prep();
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ch == next_fold && !(FD_W2P_ABLATE & 1)) {
                const float sa = ri_cur <= 2 ? 1.f : 0.f;
                const float sb = ri_cur == 0 ? 0.f : (ri_cur == 1 ? 1.f : -1.f);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float h0 = (acc[0][r] + acc[1][r]) + acc[2][r];
                    const float h1 = (acc[1][r] - acc[2][r]) - acc[3][r];
                    ya[0][r] = fmaf(sa, h0, ya[0][r]); ya[1][r] = fmaf(sa, h1, ya[1][r]);
                    yb[0][r] = fmaf(sb, h0, yb[0][r]); yb[1][r] = fmaf(sb, h1, yb[1][r]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t][r] = 0.f;
                }
            }
            if (ch == next_fold) { next_fold += cpt; ++ri_cur; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this chunk's DMAs (into the other buffer) have landed
            __builtin_amdgcn_sched_barrier(0);
            if (!(FD_W2P_ABLATE & 8)) __syncthreads();
        }
    }
    if constexpr (HALFM) {                                           // wm = 0 keeps (its own) + (wm = 1's) partial outputs
        float* red = smem + ((wn * 64) << 6) + lane;                 // [wn][64 registers][lane]; the chunk loop ended with a barrier
        if (wm == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                red[r << 6] = ya[0][r]; red[(16 + r) << 6] = ya[1][r]; red[(32 + r) << 6] = yb[0][r]; red[(48 + r) << 6] = yb[1][r];
            }
        }
        __syncthreads();
        if (wm == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            ya[0][r] += red[r << 6]; ya[1][r] += red[(16 + r) << 6]; yb[0][r] += red[(32 + r) << 6]; yb[1][r] += red[(48 + r) << 6];
        }
    }
    w2p_epilogue<STATS>(g, ya, yb, p0, m0, lim, plane2, W2, hw, lane, HALFM ? 0 : wm, wn);
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
    int slab_rows;       // rows per output channel in a slab: 9 = [ky][kx], 12 = [ri][kx] (k_wgrad_wino<.., true>)
    int xcds_per_slice;  // slice_major == 2 with fewer than 8 slices: XCDs per slice (8 / slices), else 1
    int adv_n, adv_y, adv_j;   // one chunk of WGP pairs = adv_n images + adv_y rows + adv_j pairs (host: divisions once per launch)
};
constexpr int WGP = 16;                                          // pairs per chunk (GEMM-K 16 -> 8 MFMA k-steps)
#ifndef FD_WGRAD_ABLATE       // timing experiments only (wrong results), scripts/build_ablation.sh: 1 no global loads in the loop,
#define FD_WGRAD_ABLATE 0     // 2 no LDS stores, 4 no border masks, 8 no chunk index arithmetic, 16 no barrier, 32 no operand
#endif                        // transforms, 64 no operand reads

// Both operands stay RAW in LDS and the transforms P = (y0, y0+y1, y0-y1, y1), Q = (d0-d2, d1+d2, d2-d1, d1-d3) are applied when the
// MFMA operands are read.  dY: one row of the chunk's 32 pixels per output channel (stride 34 floats: a lane (= channel) reads
// 8-byte pairs at 34 i mod 64 - 32 different bank pairs).  X: per input channel and PAIR the four pixels (d0, d1, d2, d3) the pair's
// products need, i.e. every pair carries its own left / right neighbour pixel (stride 68 floats: 16-byte reads at 4 i mod 64 banks).
// The loader thread of a pair knows whether it touches an image border and writes the padding value (0, or the mirror pixel) into
// d0 / d3 itself, so the readers need no border flags, no neighbour-cell reads and no halo cells: the round-3a layout (one raw
// 34-float row per channel, flags as scalar lane masks) spent 10 scalar + 2 vector instructions and 2 extra LDS reads per k-step on
// them.  26 KB per chunk and buffer, 52 KB per workgroup: three workgroups per CU.
constexpr int LDG = 2 * WGP + 2;                                 // dY row stride
constexpr int LDX = 4 * WGP + 4;                                 // X row stride: 16 pairs x (d0, d1, d2, d3) + 4
constexpr int WG_BUF_FLOATS = WBM * LDG + WBN * LDX;             // dY rows + X rows
constexpr int WG_LDS_FLOATS = 2 * WG_BUF_FLOATS;

// TWOD: the transposed F(2x2, 3x3) algorithm - the vertical direction is transformed as well.  The GEMM-K unit is a 2x2 tile of dY
// (image rows 2 ty, 2 ty + 1) instead of a pixel pair, and the "kernel row" of a workgroup becomes a row COMPONENT ri = 0 .. 3:
//   dY row combination  (y_r0,  y_r0 + y_r1,  y_r0 - y_r1,  y_r1)[ri]          (rows 2 ty, 2 ty + 1)
//   X  row combination  (x_r0 - x_r2,  x_r1 + x_r2,  x_r2 - x_r1,  x_r1 - x_r3)[ri]   (rows 2 ty - 1 .. 2 ty + 2, padded like the columns)
// formed by the LOADER (two row loads per operand, one fused multiply-add per value) before the pair goes to LDS; everything behind
// that - LDS layout, operand reads, the horizontal transforms, the MFMA loop, the horizontal output transform - is the 1-D kernel's.
// 4 components x half the K of 3 kernel rows: 16 products per 2x2 tile instead of 24 (direct: 36).  Slab rows are [ri][kx]; the
// vertical output transform dW[ky] = (T0 + (T1+T2)/2, (T1-T2)/2, (T1+T2)/2 - T3) is applied by k_wgrad_finish9<12> while it sums the slices.
// HALFM (round 5): at most 32 output channels (the depth decoder's upconv(1, *)) - rows 32 .. 63 of the tile do not exist, so the two waves that
// would own them (wm = 1) take the SECOND HALF OF EVERY CHUNK'S K-STEPS of the first 32 rows instead, and the two partial sums meet in LDS
// once, in front of the epilogue (fixed order: deterministic).  Without it half of the launch's matrix instructions multiply clamped rows.
template <bool REFL, bool TWOD, bool HALFM = false>      // REFL: reflection (decoder) or zero (ResNet trunk) padding - a template flag keeps the border selects out of the trunk's loop
__global__ void __launch_bounds__(WNT) k_wgrad_wino(WinoWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1;
    const int HT = TWOD ? g.H >> 1 : g.H;                        // rows of GEMM-K units per image
    constexpr int R = TWOD ? 4 : 3;                              // workgroup groups along the vertical direction
    const int plane2 = HT * W2, Np = g.Nb * plane2;              // pixel pairs / tiles: < 2^29 (size guard of the entry point)
    const unsigned hw = (unsigned)(g.H * g.W);
    // grid (default): x = (kernel row, input-channel tile), y = output-channel tile, z = pixel slice.  The alternative
    // (slice_major: x = slice with the slice count a multiple of 8, so that all workgroups of a slice share an XCD / L2) measured
    // slightly slower in the training step and is kept as a tuning switch (FD_WINO_WGRAD_MAP=1).
    const int ctiles = (g.C + WBN - 1) / WBN;
    int bt, bs, by;
    if (g.slice_major == 2) {
        // 1-D grid, XCD-aware: consecutive workgroup ids go round-robin to the 8 XCDs, so id L runs on XCD L % 8.  All (kernel row,
        // c tile, m tile) workgroups of one pixel slice get ids 8 apart - same XCD, dispatched back to back - and find the slice's
        // dY / X rows in that XCD's L2 (the 3 kernel rows alone re-read both operands: 3x the HBM traffic when they sit on 3 XCDs).
        const int mtiles = (g.M + WBM - 1) / WBM, nt = R * ctiles * mtiles;
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        int t;
        if (g.xcds_per_slice <= 1) { t = k % nt; bs = (k / nt) * 8 + xcd; }
        // fewer than 8 slices (2 or 4: ResNet layer3 at the step's batch sizes): a slice owns 8 / slices XCDs, each of which takes a
        // contiguous range of the slice's tiles (m-tile major): it reads the slice's X rows once and only its own m tiles' dY rows
        else { const int per = nt / g.xcds_per_slice; bs = xcd / g.xcds_per_slice; t = (xcd % g.xcds_per_slice) * per + k; }
        by = t / (R * ctiles);
        bt = t - by * R * ctiles;
    } else {
        bt = g.slice_major ? blockIdx.z : blockIdx.x; bs = g.slice_major ? blockIdx.x : blockIdx.z; by = blockIdx.y;
    }
    const int ky = bt / ctiles, c0 = (bt - ky * ctiles) * WBN;
    const int m0 = by * WBM;
    const int pp_lo = (int)((long)bs * g.pairs_per_split < Np ? (long)bs * g.pairs_per_split : Np);
    const int pp_hi = (long)pp_lo + g.pairs_per_split < Np ? pp_lo + (int)g.pairs_per_split : Np;
    const int nchunk = pp_hi > pp_lo ? (pp_hi - pp_lo + WGP - 1) / WGP : 0;

    // ---- loader: pair p of the chunk, rows rw + 16 i (dY rows = output channels, X rows = input channels)
    const int p = tid & 15, rw = tid >> 4;
    unsigned a_row[4], b_row[4];                                 // byte offsets of the 4 channel rows (clamped: never stored)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + rw + 16 * i; m = m < g.M ? m : g.M - 1;
        int c = c0 + rw + 16 * i; c = c < g.C ? c : g.C - 1;
        a_row[i] = 4u * (unsigned)m * hw; b_row[i] = 4u * (unsigned)c * hw;
    }
    constexpr bool refl = REFL;
    const int H2m2 = 2 * g.H - 2;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.dY);
    // X with its true size: the 16-byte load of a pair may reach one pixel past the tensor's last one - that lane reads 0.0
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    f32x2 ra[4], rb[TWOD ? 4 : 1];
    float4 rx[4], rz[TWOD ? 4 : 1];
    unsigned a_off = FD_OOB, x_off = FD_OOB, a_off2 = FD_OOB, x_off2 = FD_OOB;
    // TWOD, row component ri = ky (workgroup-uniform): which rows are combined, and the sign of the second one
    const int yr_a = ky == 3 ? 1 : 0;
    const bool y_two = ky == 1 || ky == 2;
    const float y_sgn = ky == 2 ? -1.f : 1.f;
    const int xr_a = ky == 0 ? 0 : (ky == 2 ? 2 : 1), xr_b = ky == 3 ? 3 : (ky == 2 ? 1 : 2);
    const float x_sgn = ky == 1 ? 1.f : -1.f;
    int pf = 0, rf = 0;          // bit 0 / 1: the pair of the PREPARED chunk (pf) / of the chunk whose data sit in ra, rx (rf) starts / ends an image row
    int pc = pp_lo;                                              // first pair of the chunk being prepared
    // (image, row, pair in row) of this thread's pair of the chunk being prepared: divided out once, then advanced by one chunk per
    // call with two carries - the two integer divisions per chunk of the first version were 15 % of the kernel (ablation, profiles/)
    int cn, cy, cj;
    {
        const int pq = pp_lo + p;
        cn = pq / plane2;
        const int rem = pq - cn * plane2;
        cy = rem / W2; cj = rem - cy * W2;
    }
    auto prep_chunk = [&](bool live) __attribute__((always_inline)) {
        const int pg = pc + p;
        const bool ok = live & (pg < pp_hi);
        const int n = cn, y = cy, j = cj;
        const bool e_left = j == 0, e_right = 2 * j + 2 >= g.W;
        auto x_row = [&](int r) __attribute__((always_inline)) {              // byte offset of the pair's four pixels in image row r (padded)
            const bool inb = (unsigned)r < (unsigned)g.H;
            int rr_ = r < 0 ? -r : r;
            rr_ = rr_ >= g.H ? H2m2 - rr_ : rr_;
            const int ruse = refl ? rr_ : r;
            const bool okb = ok & (refl | inb);
            const unsigned base = 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)(ruse * g.W + 2 * j));
            // four pixels from column 2j - 1 on; a pair at the left border has no such column: it loads from 2j and shifts (store_row)
            return okb ? (e_left ? base : base - 4u) : FD_OOB;
        };
        if constexpr (TWOD) {
            const unsigned ya = 4u * ((unsigned)n * (unsigned)g.M * hw + (unsigned)((2 * y + yr_a) * g.W + 2 * j));
            a_off = ok ? ya : FD_OOB;
            a_off2 = (ok & y_two) ? ya + 4u * (unsigned)g.W : FD_OOB;        // rows 2 ty and 2 ty + 1 (yr_a = 0 whenever both are used)
            x_off = x_row(2 * y - 1 + xr_a);
            x_off2 = x_row(2 * y - 1 + xr_b);
        } else {
            a_off = ok ? 4u * ((unsigned)n * (unsigned)g.M * hw + (unsigned)(y * g.W + 2 * j)) : FD_OOB;
            x_off = x_row(y + ky - 1);
        }
        pf = (e_left ? 1 : 0) | (e_right ? 2 : 0);
        pc += WGP;
        // advance (cn, cy, cj) by one chunk (values past the slice are never used: `ok` is false there)
        cj += g.adv_j;
        const bool c1 = cj >= W2;
        cj -= c1 ? W2 : 0;
        cy += g.adv_y + (c1 ? 1 : 0);
        const bool c2 = cy >= HT;
        cy -= c2 ? HT : 0;
        cn += g.adv_n + (c2 ? 1 : 0);
    };
    auto load_row = [&](int i) __attribute__((always_inline)) {
        if (!HALFM || i < 2) ra[i] = fd_ldg64(rsY, a_off + a_row[i]);     // FD_OOB + (< 2^31) stays out of range: reads 0  (HALFM: dY rows 0 .. 31 only)
        rx[i] = fd_ldg128(rsX, x_off + b_row[i]);
        if constexpr (TWOD) {
            if (!HALFM || i < 2) rb[i] = fd_ldg64(rsY, a_off2 + a_row[i]);
            rz[i] = fd_ldg128(rsX, x_off2 + b_row[i]);
        }
    };
    auto store_row = [&](int buf, int i) __attribute__((always_inline)) {
        float* qa = smem + buf * WG_BUF_FLOATS + (rw + 16 * i) * LDG + 2 * p;
        if constexpr (TWOD) {                                             // the row combinations (exact products: a +- b)
            if (!HALFM || i < 2) { ra[i].x = fmaf(y_sgn, rb[i].x, ra[i].x); ra[i].y = fmaf(y_sgn, rb[i].y, ra[i].y); }
            rx[i].x = fmaf(x_sgn, rz[i].x, rx[i].x); rx[i].y = fmaf(x_sgn, rz[i].y, rx[i].y);
            rx[i].z = fmaf(x_sgn, rz[i].z, rx[i].z); rx[i].w = fmaf(x_sgn, rz[i].w, rx[i].w);
        }
        if (!HALFM || i < 2) *reinterpret_cast<f32x2*>(qa) = ra[i];
        // (d0, d1, d2, d3) of the pair; column -1 is column 1 (reflect) or 0, column W is column W - 2 (reflect) or 0
        const bool L = rf & 1, R = rf & 2;
        float4 d;
        d.y = L ? rx[i].x : rx[i].y;
        d.z = L ? rx[i].y : rx[i].z;
        d.w = L ? rx[i].z : rx[i].w;
        d.x = L ? (refl ? d.z : 0.f) : rx[i].x;
        d.w = R ? (refl ? d.y : 0.f) : d.w;
        *reinterpret_cast<float4*>(smem + buf * WG_BUF_FLOATS + WBM * LDG + (rw + 16 * i) * LDX + 4 * p) = d;
    };

    // Wave w owns the 32 (output channels) x 32 (input channels) block (w >> 1, w & 1) of the tile with all four components (one
    // accumulator each): the output transform dW = (M0 + (M1+M2)/2, (M1-M2)/2, (M1+M2)/2 - M3) is register arithmetic.
    const int wm = wave >> 1, wn = wave & 1;
    const int arow = lane >> 5, acol = lane & 31;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int NK = WGP / 2, LS = NK / 2;
    constexpr int NKW = HALFM ? NK / 2 : NK;                     // k-steps per wave and chunk
    static_assert(!HALFM || NKW == LS, "HALFM: the four row loads / stores of a chunk sit in its four k-steps");
    const int kb = HALFM ? wm * NKW : 0;
    if (nchunk > 0) {
        prep_chunk(true);
        rf = pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(i);
#pragma unroll
        for (int i = 0; i < 4; ++i) store_row(0, i);
        prep_chunk(1 < nchunk);                                   // chunk 1: loaded now, written to LDS during chunk 0
        rf = pf;
#pragma unroll
        for (int i = 0; i < 4; ++i) load_row(i);
        prep_chunk(2 < nchunk);                                   // offsets of chunk 2, re-loaded during chunk 0
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int cur = ch & 1;
            // operands of pair k = 2 kk + arow: A = P(dY row 32 wm + acol), B = Q(X row 32 wn + acol)
            // (HALFM: every wave reads dY rows 0 .. 31; wave pair wm takes the k-steps kb .. kb + NKW - 1 of the chunk)
            const float* pa = smem + cur * WG_BUF_FLOATS + ((HALFM ? 0 : 32 * wm) + acol) * LDG + 2 * arow + 4 * kb;
            const float* pb = smem + cur * WG_BUF_FLOATS + WBM * LDG + (32 * wn + acol) * LDX + 4 * arow + 8 * kb;
            float av[2][4], bv[2][4];
            f32x2 yy;
            float4 dd;
            auto read_ops = [&](int kk2) __attribute__((always_inline)) {           // pair 2 kk2 + arow
                yy = *reinterpret_cast<const f32x2*>(pa + 4 * kk2);
                dd = *reinterpret_cast<const float4*>(pb + 8 * kk2);
            };
            auto xform = [&](int nb) __attribute__((always_inline)) {
                av[nb][0] = yy.x; av[nb][1] = yy.x + yy.y; av[nb][2] = yy.x - yy.y; av[nb][3] = yy.y;
                bv[nb][0] = dd.x - dd.z; bv[nb][1] = dd.y + dd.z; bv[nb][2] = dd.z - dd.y; bv[nb][3] = dd.y - dd.w;
            };
            read_ops(0); xform(0);
#pragma unroll
            for (int kk = 0; kk < NKW; ++kk) {
                const int cb = kk & 1, nb = cb ^ 1;
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NKW && !(FD_WGRAD_ABLATE & 64)) read_ops(kk + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < LS && !(FD_WGRAD_ABLATE & 2)) store_row(cur ^ 1, kk);              // registers loaded one chunk ago -> the other buffer
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][2], bv[cb][2], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk + 1 < NKW) { if (!(FD_WGRAD_ABLATE & 32)) xform(nb); else { for (int t = 0; t < 4; ++t) { av[nb][t] = av[cb][t]; bv[nb][t] = bv[cb][t]; } } }
                __builtin_amdgcn_sched_barrier(0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][3], bv[cb][3], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kk < LS && !(FD_WGRAD_ABLATE & 1)) load_row(kk);                        // ... and re-loaded with the chunk after next
                if (kk == (HALFM ? NKW - 1 : LS)) rf = pf;                                  // the flags travel with the registers (all four rows re-loaded by now)
                if (kk == NKW - 1 && !(FD_WGRAD_ABLATE & 8)) prep_chunk(ch + 3 < nchunk);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!(FD_WGRAD_ABLATE & 16)) __syncthreads();
        }
    }

    if constexpr (HALFM) {                                       // the two K halves of the 32 rows meet: wm = 0 keeps acc(wm = 0) + acc(wm = 1)
        float* red = smem + ((wn * 64) << 6) + lane;             // [wn][component * 16 + register][lane]; the chunk loop ended with a barrier
        if (wm == 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(t * 16 + r) << 6] = acc[t][r];
        }
        __syncthreads();
        if (wm == 1) return;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] += red[(t * 16 + r) << 6];
    }
    // ---- epilogue: slab[z][m][ky*3 + kx][c] from the four accumulators (C/D layout: column = lane & 31, row = (reg & 3) +
    //      8 * (reg >> 2) + 4 * (lane >> 5))
    const int c = c0 + 32 * wn + acol;
    // this slice's slab through a buffer resource: 32-bit offsets (a slab is M * 9 * C floats < 2^29), rows / columns past the tensor
    // are dropped by an out-of-range offset instead of a branch per row
    const __amdgpu_buffer_rsrc_t rsS = fd_make_rsrc(g.slabs + (size_t)bs * ((size_t)g.M * (3 * R) * g.C));
    const int mb = m0 + (HALFM ? 0 : 32 * wm) + 4 * arow;
    const unsigned col = (c < g.C) ? 4u * (unsigned)(ky * 3 * g.C + c) : FD_OOB;
    const unsigned row_step = 4u * (3u * R) * (unsigned)g.C, kx_step = 4u * (unsigned)g.C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const unsigned off = (m < g.M) ? col + (unsigned)m * row_step : FD_OOB;        // FD_OOB + (< 2^31) stays out of range
        const float M0 = acc[0][r], M1 = acc[1][r], M2 = acc[2][r], M3 = acc[3][r];
        const float h = 0.5f * (M1 + M2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, M0 + h), rsS, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (M1 - M2)), rsS, (int)(off + kx_step), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h - M3), rsS, (int)(off + 2u * kx_step), 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient, split precision
// k_wgrad_wino<false, true> (zero padding, transposed F(2x2, 3x3), >= 64 output channels) on the bf16 matrix pipes at fp32 accuracy
// (conv_limb.h: every fp32 operand = three bf16 limbs, six limb products, fp32 accumulation).  Same grid, slices, loader addressing, row
// combinations and epilogue; what changes is WHERE the horizontal transforms run and what LDS holds:
//   * the loader thread of a QUAD of pairs applies the vertical combination (as before), the padding (as before) AND the horizontal transforms
//     P = (y0, y0+y1, y0-y1, y1), Q = (d0-d2, d1+d2, d2-d1, d1-d3) of its four pairs, splits them into limbs and writes 8-byte pieces:
//     LDS holds, per operand, [component 4][limb 3][K half 2][row 64] 16-byte MFMA fragments of 8 pairs - the chunk's 16 pairs are ONE
//     v_mfma_f32_32x32x16_bf16 k-step;
//   * the matrix loop is 24 fragment reads + 24 MFMAs per chunk and wave (768 matrix-pipe cycles instead of the 2 048 of 32
//     v_mfma_f32_32x32x2_f32) with no vector arithmetic at all; every value is transformed and split ONCE per workgroup (the f32 kernel
//     transforms at operand-read time: once per wave that reads it);
//   * one LDS buffer (51 KB: three workgroups per CU), two barriers per chunk; the next chunk's global loads are issued in front of the
//     matrix phase and are in flight during it.
constexpr int WL_HPL = 64 * 16 + 64;              // one K half of a (component, limb) plane: 64 rows x 16 B, padded (bank spread of the two halves)
constexpr int WL_PLANE = 2 * WL_HPL;
constexpr int WL_OP = 12 * WL_PLANE;              // one operand: 4 components x 3 limbs
constexpr int WL_LDS_BYTES = 2 * WL_OP;


template <bool REFL>          // reflection (decoder) or zero (ResNet trunk) padding
__global__ void __launch_bounds__(WNT) k_wgrad_wino_limb(WinoWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned char* smemb = reinterpret_cast<unsigned char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W2 = g.W >> 1;
    const int HT = g.H >> 1;
    constexpr int R = 4;
    const int plane2 = HT * W2, Np = g.Nb * plane2;
    const unsigned hw = (unsigned)(g.H * g.W);
    const int ctiles = (g.C + WBN - 1) / WBN;
    int bt, bs, by;
    if (g.slice_major == 2) {                                    // XCD-aware 1-D grid (k_wgrad_wino)
        const int mtiles = (g.M + WBM - 1) / WBM, nt = R * ctiles * mtiles;
        const int L = blockIdx.x, xcd = L & 7, k = L >> 3;
        int t;
        if (g.xcds_per_slice <= 1) { t = k % nt; bs = (k / nt) * 8 + xcd; }
        else { const int per = nt / g.xcds_per_slice; bs = xcd / g.xcds_per_slice; t = (xcd % g.xcds_per_slice) * per + k; }
        by = t / (R * ctiles);
        bt = t - by * R * ctiles;
    } else {
        bt = g.slice_major ? blockIdx.z : blockIdx.x; bs = g.slice_major ? blockIdx.x : blockIdx.z; by = blockIdx.y;
    }
    const int ky = bt / ctiles, c0 = (bt - ky * ctiles) * WBN;
    const int m0 = by * WBM;
    const int pp_lo = (int)((long)bs * g.pairs_per_split < Np ? (long)bs * g.pairs_per_split : Np);
    const int pp_hi = (long)pp_lo + g.pairs_per_split < Np ? pp_lo + (int)g.pairs_per_split : Np;
    const int nchunk = pp_hi > pp_lo ? (pp_hi - pp_lo + WGP - 1) / WGP : 0;

    // ---- loader: thread = (row tid / 4 of both operands, QUAD tid % 4 = four consecutive pairs of the chunk, one tile row: W / 2 % 4 == 0).
    // Four adjacent lanes read 128 contiguous bytes of a dY row; two pairs of one component make one split2 (a dword = two consecutive K
    // positions), a quad an 8-byte store into the fragment - the first version (a thread = one pair of four rows, 2-byte stores: 96 LDS
    // store instructions per thread and chunk) ran at 0.76x the f32 kernel (profiles/round6_wgrad_wino_limb.log).
    const int q = tid & 3, row = tid >> 2;
    unsigned a_row, b_row;
    {
        int m = m0 + row; m = m < g.M ? m : g.M - 1;
        int c = c0 + row; c = c < g.C ? c : g.C - 1;
        a_row = 4u * (unsigned)m * hw; b_row = 4u * (unsigned)c * hw;
    }
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.dY);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.X), 0, (int)(4u * (unsigned)g.Nb * (unsigned)g.C * hw), 0x00020000);
    // (a register ring of two chunks - loads two iterations ahead - measured the same alone and in the step at 196 registers: removed)
    float4 ya[1][2], yb[1][2];                 // dY: 8 columns of the two tile rows
    float4 xa[1][2], xb[1][2];                 // X: columns 2j-1 .. 2j+6 (left edge: 2j .. 2j+7) of the two combined rows ...
    f32x2 xa2[1], xb2[1];                      // ... and 2j+7, 2j+8 (left edge: 2j+8, 2j+9)
    unsigned a_off = FD_OOB, x_off = FD_OOB, a_off2 = FD_OOB, x_off2 = FD_OOB;
    const int yr_a = ky == 3 ? 1 : 0;
    const bool y_two = ky == 1 || ky == 2;
    const float y_sgn = ky == 2 ? -1.f : 1.f;
    const int xr_a = ky == 0 ? 0 : (ky == 2 ? 2 : 1), xr_b = ky == 3 ? 3 : (ky == 2 ? 1 : 2);
    const float x_sgn = ky == 1 ? 1.f : -1.f;
    int pf = 0, rf = 0;                        // bit 0 / 1: the quad starts / ends an image row
    int pc = pp_lo;
    int cn, cy, cj;
    {
        const int pq = pp_lo + 4 * q;
        cn = pq / plane2;
        const int rem = pq - cn * plane2;
        cy = rem / W2; cj = rem - cy * W2;
    }
    auto prep_chunk = [&](bool live) __attribute__((always_inline)) {
        const int pg = pc + 4 * q;
        const bool ok = live & (pg < pp_hi);
        const int n = cn, y = cy, j = cj;
        const bool e_left = j == 0, e_right = 2 * j + 8 >= g.W;
        auto x_row = [&](int r) __attribute__((always_inline)) {
            const bool inb = (unsigned)r < (unsigned)g.H;
            int rr_ = r < 0 ? -r : r;
            rr_ = rr_ >= g.H ? 2 * g.H - 2 - rr_ : rr_;
            const int ruse = REFL ? rr_ : r;
            const bool okb = ok & (REFL | inb);
            const unsigned base = 4u * ((unsigned)n * (unsigned)g.C * hw + (unsigned)(ruse * g.W + 2 * j));
            return okb ? (e_left ? base : base - 4u) : FD_OOB;
        };
        const unsigned yo = 4u * ((unsigned)n * (unsigned)g.M * hw + (unsigned)((2 * y + yr_a) * g.W + 2 * j));
        a_off = ok ? yo : FD_OOB;
        a_off2 = (ok & y_two) ? yo + 4u * (unsigned)g.W : FD_OOB;
        x_off = x_row(2 * y - 1 + xr_a);
        x_off2 = x_row(2 * y - 1 + xr_b);
        pf = (e_left ? 1 : 0) | (e_right ? 2 : 0);
        pc += WGP;
        cj += g.adv_j;
        const bool c1 = cj >= W2;
        cj -= c1 ? W2 : 0;
        cy += g.adv_y + (c1 ? 1 : 0);
        const bool c2 = cy >= HT;
        cy -= c2 ? HT : 0;
        cn += g.adv_n + (c2 ? 1 : 0);
    };
    auto load_all = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        ya[S][0] = fd_ldg128(rsY, a_off + a_row); ya[S][1] = fd_ldg128(rsY, a_off + a_row + 16u);
        yb[S][0] = fd_ldg128(rsY, a_off2 + a_row); yb[S][1] = fd_ldg128(rsY, a_off2 + a_row + 16u);
        xa[S][0] = fd_ldg128(rsX, x_off + b_row); xa[S][1] = fd_ldg128(rsX, x_off + b_row + 16u); xa2[S] = fd_ldg64(rsX, x_off + b_row + 32u);
        xb[S][0] = fd_ldg128(rsX, x_off2 + b_row); xb[S][1] = fd_ldg128(rsX, x_off2 + b_row + 16u); xb2[S] = fd_ldg64(rsX, x_off2 + b_row + 32u);
    };
    // this quad's 8-byte slot inside the fragments of its row: K half q / 2, positions 4 (q % 2) .. + 3
    unsigned char* const slot = smemb + (q >> 1) * WL_HPL + row * 16 + 8 * (q & 1);
    auto store_all = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        // vertical combinations (exact products: a +- b)
        float Y[8], X[10];
        Y[0] = fmaf(y_sgn, yb[S][0].x, ya[S][0].x); Y[1] = fmaf(y_sgn, yb[S][0].y, ya[S][0].y); Y[2] = fmaf(y_sgn, yb[S][0].z, ya[S][0].z); Y[3] = fmaf(y_sgn, yb[S][0].w, ya[S][0].w);
        Y[4] = fmaf(y_sgn, yb[S][1].x, ya[S][1].x); Y[5] = fmaf(y_sgn, yb[S][1].y, ya[S][1].y); Y[6] = fmaf(y_sgn, yb[S][1].z, ya[S][1].z); Y[7] = fmaf(y_sgn, yb[S][1].w, ya[S][1].w);
        float r[10];
        r[0] = fmaf(x_sgn, xb[S][0].x, xa[S][0].x); r[1] = fmaf(x_sgn, xb[S][0].y, xa[S][0].y); r[2] = fmaf(x_sgn, xb[S][0].z, xa[S][0].z); r[3] = fmaf(x_sgn, xb[S][0].w, xa[S][0].w);
        r[4] = fmaf(x_sgn, xb[S][1].x, xa[S][1].x); r[5] = fmaf(x_sgn, xb[S][1].y, xa[S][1].y); r[6] = fmaf(x_sgn, xb[S][1].z, xa[S][1].z); r[7] = fmaf(x_sgn, xb[S][1].w, xa[S][1].w);
        r[8] = fmaf(x_sgn, xb2[S].x, xa2[S].x); r[9] = fmaf(x_sgn, xb2[S].y, xa2[S].y);
        // columns 2j-1 .. 2j+8; a quad at the left edge was loaded from column 2j on (shift); its column -1 and the right edge's column W
        // are the padding: 0, or the mirror columns 1 and W - 2
        const bool L = rf & 1, Rr = rf & 2;
#pragma unroll
        for (int k = 1; k < 10; ++k) X[k] = L ? r[k - 1] : r[k];
        X[0] = L ? (REFL ? X[2] : 0.f) : r[0];
        X[9] = Rr ? (REFL ? X[7] : 0.f) : X[9];
        // horizontal transforms of the four pairs, limbs, fragments (component t: planes 3 t .. 3 t + 2 = limbs h, m, l)
        unsigned char* qa = slot;
        unsigned char* qb = slot + WL_OP;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float pv[4], qv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float y0 = Y[2 * k], y1 = Y[2 * k + 1];
                const float d0 = X[2 * k], d1 = X[2 * k + 1], d2 = X[2 * k + 2], d3 = X[2 * k + 3];
                pv[k] = t == 0 ? y0 : (t == 1 ? y0 + y1 : (t == 2 ? y0 - y1 : y1));
                qv[k] = t == 0 ? d0 - d2 : (t == 1 ? d1 + d2 : (t == 2 ? d2 - d1 : d1 - d3));
            }
            unsigned h0, m0_, l0, h1, m1, l1;
            fdlimb::split2(pv[0], pv[1], h0, m0_, l0); fdlimb::split2(pv[2], pv[3], h1, m1, l1);
            *reinterpret_cast<u32x2*>(qa + (3 * t) * WL_PLANE) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(qa + (3 * t + 1) * WL_PLANE) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(qa + (3 * t + 2) * WL_PLANE) = u32x2{l0, l1};
            fdlimb::split2(qv[0], qv[1], h0, m0_, l0); fdlimb::split2(qv[2], qv[3], h1, m1, l1);
            *reinterpret_cast<u32x2*>(qb + (3 * t) * WL_PLANE) = u32x2{h0, h1};
            *reinterpret_cast<u32x2*>(qb + (3 * t + 1) * WL_PLANE) = u32x2{m0_, m1};
            *reinterpret_cast<u32x2*>(qb + (3 * t + 2) * WL_PLANE) = u32x2{l0, l1};
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int arow = lane >> 5, acol = lane & 31;
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nchunk > 0) {
        constexpr std::integral_constant<int, 0> S0{};
        prep_chunk(true);
        rf = pf;
        load_all(S0);
        prep_chunk(1 < nchunk);
        const unsigned char* fa = smemb + arow * WL_HPL + (32 * wm + acol) * 16;
        const unsigned char* fb = smemb + WL_OP + arow * WL_HPL + (32 * wn + acol) * 16;
        for (int ch = 0; ch < nchunk; ++ch) {
            store_all(S0);
            __syncthreads();
            load_all(S0);                                             // chunk ch + 1: in flight during the matrix phase
            rf = pf;
            prep_chunk(ch + 2 < nchunk);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                uint4 af[3], bf[3];
#pragma unroll
                for (int Lm = 0; Lm < 3; ++Lm) {
                    af[Lm] = *reinterpret_cast<const uint4*>(fa + (t * 3 + Lm) * WL_PLANE);
                    bf[Lm] = *reinterpret_cast<const uint4*>(fb + (t * 3 + Lm) * WL_PLANE);
                }
                FD_WLIMB_MFMA6(acc[t], af, bf);
            }
            __syncthreads();
        }
    }
    // ---- epilogue (k_wgrad_wino's): slab[z][m][ri * 3 + kx][c] from the four accumulators
    const int c = c0 + 32 * wn + acol;
    const __amdgpu_buffer_rsrc_t rsS = fd_make_rsrc(g.slabs + (size_t)bs * ((size_t)g.M * (3 * R) * g.C));
    const int mb = m0 + 32 * wm + 4 * arow;
    const unsigned col = (c < g.C) ? 4u * (unsigned)(ky * 3 * g.C + c) : FD_OOB;
    const unsigned row_step = 4u * (3u * R) * (unsigned)g.C, kx_step = 4u * (unsigned)g.C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb + (r & 3) + 8 * (r >> 2);
        const unsigned off = (m < g.M) ? col + (unsigned)m * row_step : FD_OOB;
        const float M0 = acc[0][r], M1 = acc[1][r], M2 = acc[2][r], M3 = acc[3][r];
        const float hh = 0.5f * (M1 + M2);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, M0 + hh), rsS, (int)off, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 0.5f * (M1 - M2)), rsS, (int)(off + kx_step), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hh - M3), rsS, (int)(off + 2u * kx_step), 0, 0);
    }
}

inline int wino_splits(const fd_conv_desc* d, int M, int C) {
    const long tiles = (long)fd_cdiv((long)d->N * d->H * (d->W / 2), WBN) * fd_cdiv(M, WBM);
    const int nchunk = 3 * (C / WBKC);
    int sp = 1;
    const long target = fd_tun().wino_target;              // alone on the GPU 768 is best; inside the step 256-384 (less slab traffic)
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
// F(2x2, 3x3) (k_conv_wino2d) for the layers whose matrix work dwarfs their output: Cin * Cout >= 256 * 256 (FD_WINO_FWD_2D_MIN) and
// whole 2x2 tiles.  A function of the descriptor and of fd_tuning, so that the weight-layout size, the workspace size, the
// re-layout job and the launch agree.
// -> 0: F(2, 3) along x (k_conv_wino), 1: F(2x2, 3x3) with the row components as slabs (k_conv_wino2d + k_wino2d_finish),
//    2: F(2x2, 3x3) with all 16 components in one workgroup (k_conv_wino2p): the layers with enough 2x2 tiles to fill the chip
//    without splitting anything - ResNet layer1 / layer2 at the step's batch sizes, the decoder's wide full-resolution blocks
int wino_fwd_mode(const fd_conv_desc* d) {
    if (!wino_fwd_ok(d) || d->H % 2 != 0 || (long)d->Cout * 4 * d->Cin * 4 * 4 >= 2147483648L) return 0;
    const fd_tuning& t = fd_tun();
    const long wgs = (long)fd_cdiv((long)d->N * (d->H / 2) * (d->W / 2), WBN) * fd_cdiv(d->Cout, WBM);
    // (reflect padding = a ConvBlock of the depth decoder: these run ALONE on the main stream - decoder -> loss -> decoder is the step's
    // serial section - where the stand-alone time decides, and there the slab kernel wins from 128 x 64 channels on:
    // upconv(3,1) 77 against 113 us, upconv(2,1) 85 against 98 us, scripts/decoder_conv_time.py; the trunk's zero-padded layers run
    // beside three other streams, where fewer matrix cycles per launch decide: k_conv_wino2p, -0.7 ms per step)
    const long cc_min = d->pad_mode == 1 ? (t.wino_fwd_2d_min < 8192 ? t.wino_fwd_2d_min : 8192) : t.wino_fwd_2d_min;
    const bool deep = t.wino_fwd_2d_min > 0 && (long)d->Cin * d->Cout >= cc_min;
    if (t.wino_fwd_2dp_min_wgs > 0 && wgs >= (long)t.wino_fwd_2dp_min_wgs && (!deep || t.wino_fwd_2dp_deep)) return 2;
    return deep ? 1 : 0;
}
bool wino_fwd_2d(const fd_conv_desc* d) { return wino_fwd_mode(d) != 0; }      // the weights are U2[t][m][ri][c] for both 2-D kernels
// the F(2x2, 3x3) slab kernel with a split-precision matrix loop (k_conv_wino2d_limb): its weights are the limb image of U2
bool wino_fwd_limb(const fd_conv_desc* d) {
    return fd_tun().wino_fwd_limb != 0 && wino_fwd_mode(d) == 1 && d->Cout >= 64 && d->W % 4 == 0 && d->Cin % 16 == 0 && d->Cin >= 64;
}
// channel splits of the 2-D kernel on top of its four row components
// channel splits of the 2-D slab kernels on top of their four row components, for workgroup tiles of `bm` output channels
inline int wino2d_ksplits_bm(const fd_conv_desc* d, int bm) {
    const long tiles = 4L * fd_cdiv((long)d->N * (d->H / 2) * (d->W / 2), WBN) * fd_cdiv(d->Cout, bm);
    const long target = fd_tun().wino_target;
    long ks = tiles < target ? target / tiles : 1;
    const long cap = d->Cin / WBKC / 4 > 0 ? d->Cin / WBKC / 4 : 1;          // at least 4 chunks per split
    if (ks > cap) ks = cap;
    if (ks > 4) ks = 4;
    return ks < 1 ? 1 : (int)ks;
}
// k_conv_wino2d_m128 (128 output channels per workgroup, two workgroups per CU) instead of k_conv_wino2d (64, three per CU): its
// loop carries 40 % fewer vector instructions per matrix instruction, but it halves the workgroup count.  Stand-alone it wins exactly
// where its launch fills the chip's workgroup slots better (scripts/conv2d_m128_time.py: layer3 / layer4 at batch 12 and layer4 at
// batch 24 -4 ... -10 %; layer3 at batch 24, 720 -> 360 workgroups, +14 %) - fd_tuning.wino_fwd_2d_m128 = 2 chooses by that rule.
// INSIDE the training step the other streams fill a launch's empty slots, and the kernel with the leaner loop is the better one
// everywhere it can run (19.07 / 19.07 / 19.17 / 19.14 ms against 19.15 - 19.34 for the rule or the 64-channel kernel): the default (1).
// The launcher additionally needs a 16-byte aligned x.
inline bool wino2d_m128(const fd_conv_desc* d) {
    if (fd_tun().wino_fwd_2d_m128 == 0 || d->Cout % M2_BM != 0 || d->W % 4 != 0 || d->Cin % M2_KC != 0) return false;
    if (fd_tun().wino_fwd_2d_m128 != 2) return true;
    const long px = fd_cdiv((long)d->N * (d->H / 2) * (d->W / 2), WBN);
    const long n64 = 4L * px * fd_cdiv(d->Cout, WBM) * wino2d_ksplits_bm(d, WBM), n128 = 4L * px * (d->Cout / M2_BM) * wino2d_ksplits_bm(d, M2_BM);
    const long s64 = 3 * 256, s128 = 2 * 256;                               // workgroup slots of the chip
    // fill = n / (rounds * slots), compared as cross products
    return n128 * (fd_cdiv(n64, s64) * s64) > n64 * (fd_cdiv(n128, s128) * s128);
}
inline int wino2d_ksplits(const fd_conv_desc* d) { return wino2d_ksplits_bm(d, (!wino_fwd_limb(d) && wino2d_m128(d)) ? M2_BM : WBM); }
long wino_wt_floats(const fd_conv_desc* d) {
    if (wino_fwd_limb(d)) return 24L * d->Cout * d->Cin;                  // 16 components x 3 bf16 limbs
    return 4L * d->Cout * (wino_fwd_2d(d) ? 4 : 3) * d->Cin;
}
long wino_ws_floats(const fd_conv_desc* d) {
    const int mode = wino_fwd_mode(d);
    if (mode == 2) return 0;
    if (mode == 1) return 4L * wino2d_ksplits(d) * d->N * d->Cout * (d->H / 2) * d->W;
    const int sp = wino_splits(d, d->Cout, d->Cin);
    return sp > 1 ? (long)sp * d->N * d->Cout * d->H * d->W : 0;
}
// U for the convolution `d` computes (for a data gradient: Cin / Cout already swapped, flip = 1; w is always [Cout][Cin][3][3] of the layer)
int wino_weight_launch(const fd_conv_desc* d, const float* w, float* U, int flip, hipStream_t st) {
    const int M = d->Cout, C = d->Cin;
    const bool twod = wino_fwd_2d(d);
    if (wino_fwd_limb(d)) {
        const long nl = (long)M * 4 * (C >> 3);
        hipLaunchKernelGGL(k_wino_weight2d_limb, dim3(fd_cdiv(nl, 256) > 4096 ? 4096 : fd_cdiv(nl, 256)), dim3(256), 0, st, w, reinterpret_cast<uint4*>(U), M, C, flip);
        FD_LAUNCH_CHECK("wino weight transform (limbs)");
        return 0;
    }
    const long n = (long)M * (twod ? 4 : 3) * C;
    const dim3 grid(fd_cdiv(n, 256) > 4096 ? 4096 : fd_cdiv(n, 256));
    if (twod) hipLaunchKernelGGL(k_wino_weight2d, grid, dim3(256), 0, st, w, U, M, C, flip);
    else hipLaunchKernelGGL(k_wino_weight, grid, dim3(256), 0, st, w, U, M, C, flip);
    FD_LAUNCH_CHECK("wino weight transform");
    return 0;
}
// y = act(conv3x3(x; U) + bias); d describes the convolution being computed (for a data gradient: Cin / Cout already swapped).
// slots of BatchNorm partial sums per (image, channel) the kernel can emit for `d`, 0 if not (split-K, tiles across images)
int wino_stat_slots(const fd_conv_desc* d) {
    if (!wino_fwd_ok(d) || d->act != 0) return 0;
    const int mode = wino_fwd_mode(d);
    if (mode == 1) return 0;
    if (mode == 2) {                                                       // slots of 32 tiles x 4 pixels
        const long tiles = (long)(d->H / 2) * (d->W / 2);
        if (tiles % WBN == 0) return (int)(2 * tiles / WBN);
        // half a tile left over per image (ResNet layer2 at 640x192: 480 tiles): the direct-to-LDS kernel tiles image by image
        return (tiles % 32 == 0 && d->W % 4 == 0 && fd_tun().wino_fwd_2dp_dma != 0) ? (int)(tiles / 32) : 0;
    }
    const long plane2 = (long)d->H * (d->W / 2);
    if (plane2 % WBN != 0 || wino_splits(d, d->Cout, d->Cin) != 1) return 0;
    return (int)(2 * plane2 / WBN);
}

bool wino_fwd_slab_route(const fd_conv_desc* d) { return wino_fwd_ok(d) && wino_fwd_mode(d) == 1; }

int wino_conv_launch(const fd_conv_desc* d, const float* x, const float* U, const float* bias, float* y, float* ws, hipStream_t st,
                     const float* add, float* stat_part, const BnAfterConv* bn) {
    if (bn && (wino_fwd_mode(d) != 1 || bias || add || d->act != 0)) { fd_set_error("wino conv: the fused BatchNorm needs the slab route without bias / activation"); return -1; }
    WinoArgs g = {};
    g.U = U; g.X = x; g.Y = y; g.bias = bias; g.slabs = ws; g.add = add;
    g.stat_part = stat_part; g.stat_slots = stat_part ? wino_stat_slots(d) : 0;
    if (stat_part && g.stat_slots == 0) { fd_set_error("wino conv: no statistics epilogue for this shape"); return -1; }
    g.M = d->Cout; g.C = d->Cin; g.Nb = d->N; g.H = d->H; g.W = d->W;
    g.pad_mode = d->pad_mode; g.act = d->act;
    const long out_total = (long)d->N * d->Cout * d->H * d->W;
    g.slab_stride = out_total;
    const int mode = wino_fwd_mode(d);
    const bool twod = mode == 1;
    const int sp = twod ? 4 * wino2d_ksplits(d) : (mode == 2 ? 1 : wino_splits(d, d->Cout, d->Cin));
    if (sp > 1 && !ws) { fd_set_error("wino conv: split-K workspace missing"); return -1; }
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2d), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2p<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2p<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2p_dma<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2p_dma<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    if (mode == 2) {
        const long img_tiles = (long)(d->H / 2) * (d->W / 2);
        const bool aligned = stat_part && img_tiles % WBN != 0;             // statistics on a plane of 64 k + 32 tiles: tile image by image
        const int gx2 = aligned ? d->N * fd_cdiv(img_tiles, WBN) : fd_cdiv((long)d->N * img_tiles, WBN), gy2 = fd_cdiv(d->Cout, WBM);
        g.img_tiles = aligned ? fd_cdiv(img_tiles, WBN) : 0;
        g.xcd_swizzle = (gx2 % 8 == 0 && gx2 >= 16) ? 1 : 0;
        // direct-to-LDS activations need 16-byte pieces that stay inside one image row and a 16-byte aligned tensor
        const bool vdma = fd_tun().wino_fwd_2dp_dma != 0 && d->W % 4 == 0 && ((uintptr_t)x & 15) == 0;
        if (aligned && !vdma) { fd_set_error("wino conv: the statistics epilogue of this shape needs a 16-byte aligned input"); return -1; }
        if (vdma) {
            if (stat_part) hipLaunchKernelGGL(k_conv_wino2p_dma<true>, dim3(gx2, gy2), dim3(WNT), sizeof(float) * W2D_LDS_FLOATS, st, g);
            else if (d->Cout <= 32 && fd_tun().wino_fwd_halfm != 0) {          // the decoder's 32-channel blocks: both wave pairs on rows 0 .. 31
                static FdLdsAttrOnce attr_h;
                if (attr_h.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2p_dma<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_h.mark(); }
                hipLaunchKernelGGL((k_conv_wino2p_dma<false, true>), dim3(gx2, gy2), dim3(WNT), sizeof(float) * W2D_LDS_FLOATS, st, g);
            }
            else hipLaunchKernelGGL(k_conv_wino2p_dma<false>, dim3(gx2, gy2), dim3(WNT), sizeof(float) * W2D_LDS_FLOATS, st, g);
        } else if (stat_part) hipLaunchKernelGGL(k_conv_wino2p<true>, dim3(gx2, gy2), dim3(WNT), sizeof(float) * W_LDS_FLOATS, st, g);
        else hipLaunchKernelGGL(k_conv_wino2p<false>, dim3(gx2, gy2), dim3(WNT), sizeof(float) * W_LDS_FLOATS, st, g);
        FD_LAUNCH_CHECK("k_conv_wino2p");
        return 0;
    }
    if (twod) {
        if (stat_part) { fd_set_error("wino conv: no statistics epilogue for this shape"); return -1; }
        const int HT = d->H / 2;
        const int gx2 = fd_cdiv((long)d->N * HT * (d->W / 2), WBN);
        g.slab_stride = out_total / 2;                                     // S_ri: [N][M][H/2][W]
        const bool limb2d = wino_fwd_limb(d);                              // the weights are the limb image: only k_conv_wino2d_limb reads it
        if (limb2d && ((uintptr_t)x & 15) != 0) { fd_set_error("wino conv: the split-precision slab kernel needs a 16-byte aligned input"); return -1; }
        const bool m128 = !limb2d && wino2d_m128(d) && ((uintptr_t)x & 15) == 0;      // (unaligned x: k_conv_wino2d with the same split count)
        const int gy2 = fd_cdiv(d->Cout, m128 ? M2_BM : WBM);
        const int xmap = 1;        // XCD-aware 1-D grid (plain 3-D grid: layer4 162 instead of 79 MB of HBM traffic per launch, -0.35 % in the step)
        g.xcd_swizzle = (gx2 % 8 == 0 && gx2 >= 16) ? 1 : 0;
        dim3 grid(gx2, gy2, sp);
        if (xmap && (gy2 * sp) % 8 == 0) { g.xcd_swizzle = 2; g.gx = gx2; g.gy = gy2; g.gz = sp; grid = dim3((unsigned)(gx2 * gy2 * sp)); }
        if (limb2d) {
            static FdLdsAttrOnce attr_l;
            if (attr_l.needed()) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv_wino2d_limb), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_l.mark(); }
            hipLaunchKernelGGL(k_conv_wino2d_limb, grid, dim3(WNT), (size_t)W2L_LDS_BYTES, st, g);
        }
        else if (m128) hipLaunchKernelGGL(k_conv_wino2d_m128, grid, dim3(WNT), sizeof(float) * M2_LDS_FLOATS, st, g);
        else hipLaunchKernelGGL(k_conv_wino2d, grid, dim3(WNT), sizeof(float) * W_LDS_FLOATS, st, g);
        FD_LAUNCH_CHECK("k_conv_wino2d");
        if (bn)             // the slab reduction + vertical output transform inside the small-plane BatchNorm kernel that follows (round 5)
            return bn_small_slabs_launch(ws, g.slab_stride, sp / 4, y, *bn, d->N, d->Cout, d->H, d->W, st);
        const unsigned total2 = (unsigned)(out_total / 4);               // one thread per (tile row, column pair)
        const unsigned blocks = (total2 + 255u) / 256u;
        hipLaunchKernelGGL(k_wino2d_finish, dim3(blocks > 4096u ? 4096u : blocks), dim3(256), 0, st, ws, y, bias, add, total2,
                           g.slab_stride, sp / 4, HT, d->W, d->Cout, d->act);
        FD_LAUNCH_CHECK("k_wino2d_finish");
        return 0;
    }
    const int gx = fd_cdiv((long)d->N * d->H * (d->W / 2), WBN), gy = fd_cdiv(d->Cout, WBM);
    g.xcd_swizzle = (gx % 8 == 0 && gx >= 16) ? 1 : 0;
    // direct-to-LDS activations need 16-byte pieces that stay inside one image row and a 16-byte aligned tensor (otherwise the
    // register-staged loader: same results bit for bit, +0.12 ... +0.20 ms per step when forced)
    const bool vdma = d->W % 4 == 0 && ((uintptr_t)x & 15) == 0;
    const bool stats = g.stat_part != nullptr;
    auto kern = vdma ? (stats ? k_conv_wino<true, true> : k_conv_wino<true, false>) : (stats ? k_conv_wino<false, true> : k_conv_wino<false, false>);
    hipLaunchKernelGGL(kern, dim3(gx, gy, sp), dim3(WNT), sizeof(float) * W_LDS_FLOATS, st, g);
    FD_LAUNCH_CHECK("k_conv_wino");
    if (sp > 1) return fast_splitk_finish_launch(ws, y, bias, out_total, out_total, sp, (long)d->H * d->W, d->Cout, d->act, st, add);
    return 0;
}

// ---- probe entry points (scripts/wino_probe.py, tests): the Winograd path on its own
extern "C" long fd_conv3x3_wino_wt_floats(const fd_conv_desc* d) { return (d && wino_fwd_ok(d)) ? wino_wt_floats(d) : 0; }
extern "C" long fd_conv3x3_wino_ws_floats(const fd_conv_desc* d) { return (d && wino_fwd_ok(d)) ? wino_ws_floats(d) : 0; }
extern "C" int fd_conv3x3_wino_fwd(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, float* wt,
                                   int wt_ready, float* ws, void* stream) {
    FD_REQUIRE(d && x && w && y && wt, "fd_conv3x3_wino_fwd: NULL argument");
    FD_REQUIRE(wino_fwd_ok(d), "fd_conv3x3_wino_fwd: needs a 3x3 stride-1 pad-1 convolution with Cin %% 16 == 0 and an even width");
    hipStream_t st = (hipStream_t)stream;
    if (!wt_ready)
        if (int rc = wino_weight_launch(d, w, wt, 0, st)) return rc;
    return wino_conv_launch(d, x, wt, bias, y, ws, st);
}

// ---- weight gradient
bool wino_wgrad_ok(const fd_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->Cin % 16 == 0 && d->Cin >= 64 && d->Cout >= fd_tun().wino_wgrad_min_cout &&
           d->W % 2 == 0 && !d->in_norm;
}
// The 2-D algorithm needs whole 2x2 tiles of dY (fd_tuning.wino_wgrad_2d = 0: the 1-D kernel everywhere, for A/B timing)
bool wino_wgrad_2d(const fd_conv_desc* d) {
    // wino_wgrad_2d = 1: where Cin is a multiple of 32 (rounds 4-5); 2: every Cin the Winograd path takes (multiples of 16: the
    // Refiner decoder's 272 / 144 / 112-channel layers)
    const int mode = fd_tun().wino_wgrad_2d;
    return mode != 0 && d->H % 2 == 0 && (d->Cin % 32 == 0 || mode >= 2);
}
int wino_wgrad_splits(const fd_conv_desc* d) {
    const bool twod = wino_wgrad_2d(d);
    const long tiles = (twod ? 4L : 3L) * fd_cdiv(d->Cin, WBN) * fd_cdiv(d->Cout, WBM);
    const long Np = (long)d->N * (twod ? d->H / 2 : d->H) * (d->W / 2);
    const long target = fd_tun().wino_wgrad_target;          // in-step optimum 384 (768: -1 %)
    long sp = target / tiles;
    const long maxs = (Np + 4 * WGP - 1) / (4 * WGP);          // at least 4 chunks per split
    if (sp > maxs) sp = maxs;
    if (sp > 512) sp = 512;
    if (sp >= 8) sp &= ~7L;                                     // XCD alignment, see k_wgrad_wino
    if (sp < 1) sp = 1;
    return (int)sp;
}
long wino_wgrad_ws_floats(const fd_conv_desc* d) { return (long)wino_wgrad_splits(d) * d->Cout * (wino_wgrad_2d(d) ? 12 : 9) * d->Cin; }
int wino_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate, hipStream_t st) {
    WinoWgradArgs g = {};
    g.dY = gy; g.X = x; g.slabs = ws;
    g.M = d->Cout; g.C = d->Cin; g.Nb = d->N; g.H = d->H; g.W = d->W; g.pad_mode = d->pad_mode;
    const int sp = wino_wgrad_splits(d);
    const bool twod = wino_wgrad_2d(d);
    const int HT = twod ? d->H / 2 : d->H;
    g.slab_rows = twod ? 12 : 9;
    const long Np = (long)d->N * HT * (d->W / 2);
    long pps = (Np + sp - 1) / sp;
    pps = (pps + WGP - 1) / WGP * WGP;
    g.pairs_per_split = pps;
    {
        const int W2 = d->W / 2, plane2 = HT * W2;
        g.adv_n = WGP / plane2;
        const int rem = WGP - g.adv_n * plane2;
        g.adv_y = rem / W2; g.adv_j = rem - g.adv_y * W2;
    }
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    const int slice_major = 2;        // XCD-aware 1-D grid: layer1 HBM traffic 157 -> 76 MB per launch, step -0.5 %; (1: slice-major 3-D grid measured slower than 0)
    g.slice_major = slice_major;
    const int nt = (twod ? 4 : 3) * fd_cdiv(d->Cin, WBN);
    const int mt = fd_cdiv(d->Cout, WBM);
    g.xcds_per_slice = 1;
    if (slice_major == 2 && sp % 8 != 0) {                                // the XCD map needs whole groups of 8 slices ...
        const int q = (sp == 2 || sp == 4) ? 8 / sp : 0;                  // ... or 2 / 4 slices that own 4 / 2 XCDs each
        if (q > 0 && (nt * mt) % q == 0 && fd_tun().wino_wgrad_xcd_few != 0) g.xcds_per_slice = q;
        else g.slice_major = 0;
    }
    const dim3 grid = g.slice_major == 2 ? dim3((unsigned)(nt * mt * sp)) : (g.slice_major ? dim3(sp, mt, nt) : dim3(nt, mt, sp));
    const size_t lds = sizeof(float) * WG_LDS_FLOATS;
    const bool halfm = d->Cout <= 32 && fd_tun().wino_wgrad_halfm != 0;   // at most 32 output channels: two waves per K half (k_wgrad_wino<.., HALFM>)
    if (fd_tun().wino_wgrad_limb != 0 && twod && (d->pad_mode == 0 || fd_tun().wino_wgrad_limb >= 2) && d->Cout >= 64 && d->W % 8 == 0) {
        // split-precision matrix loop: the ResNet trunk's layers (wino_wgrad_limb = 1), the reflect-padded decoder blocks as well (2)
        static FdLdsAttrOnce attr_l;
        if (attr_l.needed()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino_limb<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino_limb<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_l.mark();
        }
        if (d->pad_mode == 1) hipLaunchKernelGGL(k_wgrad_wino_limb<true>, grid, dim3(WNT), (size_t)WL_LDS_BYTES, st, g);
        else hipLaunchKernelGGL(k_wgrad_wino_limb<false>, grid, dim3(WNT), (size_t)WL_LDS_BYTES, st, g);
    } else if (halfm) {
        static FdLdsAttrOnce attr_h;
        if (attr_h.needed()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_wino<true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr_h.mark();
        }
        if (twod) {
            if (d->pad_mode == 1) hipLaunchKernelGGL((k_wgrad_wino<true, true, true>), grid, dim3(WNT), lds, st, g);
            else hipLaunchKernelGGL((k_wgrad_wino<false, true, true>), grid, dim3(WNT), lds, st, g);
        } else {
            if (d->pad_mode == 1) hipLaunchKernelGGL((k_wgrad_wino<true, false, true>), grid, dim3(WNT), lds, st, g);
            else hipLaunchKernelGGL((k_wgrad_wino<false, false, true>), grid, dim3(WNT), lds, st, g);
        }
    } else if (twod) {
        if (d->pad_mode == 1) hipLaunchKernelGGL((k_wgrad_wino<true, true>), grid, dim3(WNT), lds, st, g);
        else hipLaunchKernelGGL((k_wgrad_wino<false, true>), grid, dim3(WNT), lds, st, g);
    } else {
        if (d->pad_mode == 1) hipLaunchKernelGGL((k_wgrad_wino<true, false>), grid, dim3(WNT), lds, st, g);
        else hipLaunchKernelGGL((k_wgrad_wino<false, false>), grid, dim3(WNT), lds, st, g);
    }
    FD_LAUNCH_CHECK("k_wgrad_wino");
    return fast_wgrad_finish_launch(ws, gw, d->Cout, d->Cin, twod ? 12 : 9, sp, accumulate, st);
}

