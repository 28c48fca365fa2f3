// 3x3 stride-1 pad-1 convolutions of the decoder's full-resolution blocks: 16 or 32 output channels, 16 or 32 input channels
// (depth_decoder.py:44-57: upconv(0,0) 32 -> 16 at 96x320, upconv(0,1) 16 -> 16 at 192x640, and their data gradients), on
// v_mfma_f32_16x16x4_f32.
//
// These layers are the tail of the step's serial chain and the implicit-GEMM kernel (conv_fast.hip) is a poor fit for them: its
// 32-row MFMA tile is half padding at 16 output channels, it stages every input pixel nine times (once per tap, 16 channels of one
// tap per chunk, a barrier per chunk), and with nine short chunks per tile the per-tile latencies dominate - 190 us for 16 -> 16 at
// 192x640, batch 12, against 43 us of matrix work and ~40 us of HBM traffic.  Here:
//   * a workgroup owns TR rows x 64 columns of one image; the input patch ((TR + 2) x 66 pixels x C channels, padding resolved by
//     the loader: mirror pixel or 0.0) is staged in LDS ONCE, one barrier per tile;
//   * all 9 x C x M weights live in registers for the whole kernel (A operand of the 16x16x4 MFMA: lane = (output channel,
//     channel-of-four), one register per (tap, channel group, 16-channel block));
//   * the B operand of a tap is the patch read at a shifted address: one ds_read_b32 per MFMA, channel stride = 16 (mod 64) banks
//     so that the four channels of a group never collide with the 16 pixels of a block;
//   * bias + activation in the epilogue, stores through a buffer resource (columns / rows past the image are dropped by the range
//     check).
// The data gradient of such a layer is the same kernel on dY with the weights read transposed and flipped (`flip`), zero padding.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include <stdlib.h>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// sigmoid / tanh: one out-of-line copy.  ELU - the activation of every layer this kernel serves - is a template case and stays
// inline: a call in the epilogue made the compiler save the 36 .. 72 weight registers around each of its 16 .. 32 invocations.
__device__ __attribute__((noinline)) float n16_act_slow(float v, int act) {
    if (act == 3) return 1.0f / (1.0f + expf(-v));
    return tanhf(v);
}
template <int ACT>      // 0 none, 1 ReLU, 2 ELU, -1: g.act at run time (sigmoid, tanh)
__device__ __forceinline__ float n16_act(float v, int act) {
    if (ACT == 0) return v;
    if (ACT == 1) return fmaxf(v, 0.f);
    if (ACT == 2) return v > 0.f ? v : expm1f(v);
    return n16_act_slow(v, act);
}

struct N16Args {
    const float* X; const float* Wt; const float* bias; float* Y;
    int Nb, H, W;
    int pad_mode, act, flip;
    int tiles_x, tiles_y;
};

constexpr int N16_COLS = 64, N16_LDW = 66;
// channel stride of the LDS patch: (TR + 2) rows of 66, rounded up to 16 (mod 64) floats
__host__ __device__ constexpr int n16_cs(int TR) {
    int cs = (TR + 2) * N16_LDW;
    while (cs % 64 != 16) ++cs;
    return cs;
}

// CG: input channels / 4, MB: output channels / 16, TR: rows per workgroup (multiple of 4)
template <int CG, int MB, int TR, int ACT>
__global__ void __launch_bounds__(256) k_conv3x3_n16(N16Args g) {
    constexpr int C = 4 * CG, M = 16 * MB, PR = TR + 2, CS = n16_cs(TR);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int b = blockIdx.x;
    const int tx = b % g.tiles_x; b /= g.tiles_x;
    const int ty = b % g.tiles_y;
    const int n = b / g.tiles_y;
    const int x0 = tx * N16_COLS, y0 = ty * TR;
    const unsigned hw = (unsigned)(g.H * g.W);
    const bool refl = g.pad_mode == 1;
    const __amdgpu_buffer_rsrc_t rsX = fd_make_rsrc(g.X);

    // ---- weights -> LDS (coalesced; a gather of 36 .. 72 single floats per lane straight from memory cost each of the 2 880
    //      workgroups ~4 us of address-unit time) -> registers.  A operand of v_mfma_f32_16x16x4_f32: lane l holds
    //      A[row l & 15][k = l >> 4].
    for (int i = tid; i < M * C * 9; i += 256) smem[i] = g.Wt[i];
    __syncthreads();
    float areg[9][CG][MB];
    {
        const int am = lane & 15, ak = lane >> 4;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const int m = 16 * mb + am, c = 4 * cg + ak;
                    // forward: W[m][c][t]; data gradient: the layer's W[c][m] with the taps flipped
                    const unsigned idx = g.flip ? (unsigned)((c * M + m) * 9 + (8 - t)) : (unsigned)((m * C + c) * 9 + t);
                    areg[t][cg][mb] = smem[idx];
                }
    }
    __syncthreads();                                     // the patch overwrites the weights' staging area

    // ---- patch -> LDS.  One line (channel, patch row) per wave and iteration: lane l fetches interior column x0 + l, lanes 0 / 1
    //      the halo columns x0 - 1 / x0 + 64; the line's base is wave-uniform (scalar offset), the column offsets are fixed.
    unsigned xo_in, xo_halo;
    {
        const int x = x0 + lane;
        // (a tile that crosses the right border holds the first padding column as an interior lane)
        xo_in = x < g.W ? 4u * (unsigned)x : ((x == g.W && refl) ? 4u * (unsigned)(g.W - 2) : FD_OOB);
        int xh = lane == 0 ? x0 - 1 : x0 + N16_COLS;
        bool ok = lane < 2;
        if (xh < 0) { ok = ok && refl; xh = 1; }
        else if (xh >= g.W) { ok = ok && refl && xh == g.W; xh = g.W - 2; }      // beyond the first padding column: feeds dropped outputs only
        xo_halo = ok ? 4u * (unsigned)xh : FD_OOB;
    }
    constexpr int LINES = C * PR;
    constexpr int LPW = (LINES + 3) / 4;                 // lines per wave
    constexpr int BATCH = (LPW + 2) / 3;                 // three rounds of loads, each with all of its lines in flight
#pragma unroll 1
    for (int i0 = 0; i0 < LPW; i0 += BATCH) {
        float vin[BATCH], vh[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int line = wave + 4 * (i0 + j);
            const int c = line / PR, r = line - c * PR;
            int y = y0 - 1 + r;
            bool ok = line < LINES;
            if (y < 0) { ok = ok && refl; y = 1; }
            else if (y >= g.H) { ok = ok && refl && y == g.H; y = g.H - 2; }
            const unsigned base = 4u * (((unsigned)n * C + (unsigned)c) * hw + (unsigned)y * (unsigned)g.W);   // wave-uniform, < 2^31
            // (base + FD_OOB stays out of range and reads 0.0; FD_OOB + FD_OOB would wrap to 0: a padding row takes the select)
            vin[j] = fd_ldg32(rsX, ok ? base + xo_in : FD_OOB);
            vh[j] = fd_ldg32(rsX, ok ? base + xo_halo : FD_OOB);
        }
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int line = wave + 4 * (i0 + j);
            if (line < LINES) {
                const int c = line / PR, r = line - c * PR;
                float* q = smem + c * CS + r * N16_LDW;
                q[1 + lane] = vin[j];
                if (lane < 2) q[lane == 0 ? 0 : N16_LDW - 1] = vh[j];
            }
        }
    }
    __syncthreads();

    // ---- MFMA: wave w owns rows w, w + 4, ... of the tile; per row 4 blocks of 16 pixels x MB blocks of 16 channels.
    //      B operand: lane l holds B[k = l >> 4][pixel l & 15] = patch[channel 4 cg + k][row + ky][16 blk + pixel + kx].
    const int bk = lane >> 4, bp = lane & 15;
    const float* pb = smem + bk * CS + bp;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(g.Y);
    float bias_r[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) bias_r[mb][r] = g.bias ? g.bias[16 * mb + 4 * bk + r] : 0.f;
#pragma unroll 1
    for (int row = wave; row < TR; row += 4) {
        f32x4 acc[4][MB];
#pragma unroll
        for (int blk = 0; blk < 4; ++blk)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[blk][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* pr = pb + row * N16_LDW;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                float bv[4];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk) bv[blk] = pr[4 * cg * CS + ky * N16_LDW + 16 * blk + kx];
#pragma unroll
                for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[blk][mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[t][cg][mb], bv[blk], acc[blk][mb], 0, 0, 0);
            }
        }
        // epilogue of the row: D[row 4 (l >> 4) + r][column l & 15] of each block
        const int y = y0 + row;
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            const int x = x0 + 16 * blk + bp;
            const bool ok = x < g.W && y < g.H;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * mb + 4 * bk + r;
                    float v = acc[blk][mb][r] + bias_r[mb][r];
                    v = n16_act<ACT>(v, g.act);
                    const unsigned off = ok ? 4u * (((unsigned)n * M + (unsigned)m) * hw + (unsigned)(y * g.W + x)) : FD_OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, (int)off, 0, 0);
                }
        }
    }
}

template <int CG, int MB, int TR, int ACT>
int n16_go2(const N16Args& a, hipStream_t st) {
    N16Args g = a;
    g.tiles_x = fd_cdiv(a.W, N16_COLS); g.tiles_y = fd_cdiv(a.H, TR);
    static_assert((4 * CG) * n16_cs(TR) >= 16 * MB * 4 * CG * 9, "the patch area also stages the weights");
    const size_t lds = sizeof(float) * (size_t)(4 * CG) * n16_cs(TR);
    auto kern = k_conv3x3_n16<CG, MB, TR, ACT>;
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)((long)g.tiles_x * g.tiles_y * a.Nb)), dim3(256), lds, st, g);
    FD_LAUNCH_CHECK("k_conv3x3_n16");
    return 0;
}
template <int CG, int MB, int TR>
int n16_go(const N16Args& a, hipStream_t st) {
    if (a.act == 0) return n16_go2<CG, MB, TR, 0>(a, st);
    if (a.act == 2) return n16_go2<CG, MB, TR, 2>(a, st);
    if (a.act == 1) return n16_go2<CG, MB, TR, 1>(a, st);
    return n16_go2<CG, MB, TR, -1>(a, st);
}
}  // namespace

// M output / C input channels of the convolution being computed (for a data gradient: the layer's Cin / Cout)
bool n16_shape_ok(const fd_conv_desc* d, int M, int C) {
    const long min_px = fd_tun().conv_n16_min_pixels;        // default 16384 - planes below 128 x 128: the patch halo and the tile quantisation eat the gain
    if (min_px < 0) return false;                            // these layers stay on the implicit-GEMM kernel (A/B timing, tests)
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->in_norm && d->H >= 2 && d->W >= 2 &&
           (long)d->H * d->W >= min_px && ((M == 16 && (C == 16 || C == 32)) || (M == 32 && C == 16)) &&
           (long)d->N * (M > C ? M : C) * d->H * d->W < (1L << 29);
}

int n16_launch(const fd_conv_desc* d, int M, int C, const float* x, const float* w, const float* bias, float* y, int flip,
               int pad_mode, int act, hipStream_t st) {
    N16Args a = {};
    a.X = x; a.Wt = w; a.bias = bias; a.Y = y;
    a.Nb = d->N; a.H = d->H; a.W = d->W; a.pad_mode = pad_mode; a.act = act; a.flip = flip;
    if (M == 16 && C == 16) return n16_go<4, 1, 8>(a, st);
    if (M == 16 && C == 32) return n16_go<8, 1, 4>(a, st);
    if (M == 32 && C == 16) return n16_go<4, 2, 8>(a, st);
    fd_set_error("n16_launch: unsupported channel counts %d -> %d", C, M);
    return -1;
}
