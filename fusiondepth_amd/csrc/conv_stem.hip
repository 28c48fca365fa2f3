// 7x7 stride-2 pad-3 stem convolution of the ResNet encoders, forward (networks/resnet_encoder.py:95 `self.encoder.conv1` with
// 2 / 3 / 4 / 6 input channels: LiDAR 2-channel map, RGB, and their frame pairs for the pose encoders; 64 output channels, no bias).
//
// Until round 3 this ran on the generic gather GEMM (k_gather_gemm<7, 7, ...>): every one of the 49 C taps of every output pixel
// was an address computation + a global load in the K loop, 64 - 68 TFLOP/s (0.41 - 0.43 of the fp32 MFMA peak) for 3 % of the
// step's flops and 2.9 % of its kernel time.  Here the operands of a whole output tile are staged ONCE:
//   * a workgroup (8 waves) owns 8 output rows x 64 output columns of one image, all 64 output channels; wave w = output row w;
//   * the (2 * 8 + 5) x (2 * 64 + 5) x C input patch goes to LDS once, split into its even and odd columns, so that tap kx of
//     output column px is element px + (kx >> 1) of plane (kx & 1): unit stride across the lanes, no bank conflicts despite the
//     stride-2 convolution; padding is resolved by the loader (zeros), the tap loop has no bounds logic at all;
//   * all 49 C x 64 weights go to LDS once ([k][m], row stride 65: the coalesced read of the OIHW array scatters into 16 banks);
//   * GEMM-K pairs two input CHANNELS (k-step = (channel pair, ky, kx), MFMA k index = channel parity), so the two half-waves of
//     an MFMA operand differ by one constant LDS offset and every operand read is `base + compile-time immediate`:
//     v_mfma_f32_32x32x2_f32, 64 channels x 64 pixels per wave = 2 x 2 tiles, 4 LDS reads per 4 MFMAs, nothing else in the loop.
//     An odd channel count (RGB) is padded to the next pair with zero weights (+33 % MFMA work on the smallest of the four stems).
// Algorithmic work 2 * 64 * 49 * C flop per output pixel; LDS: 49 * 2 CP * 65 * 4 (weights) + 2 CP * 21 * 2 * 68 * 4 (patch)
// = 76 + 69 KB for C = 6: one workgroup of 8 waves per CU.
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"

namespace {

constexpr int ST_TR = 8, ST_TC = 64, ST_NT = 512;
constexpr int ST_PR = 2 * ST_TR + 5;            // patch rows
constexpr int ST_PW = 2 * ST_TC + 5;            // patch columns (133)
constexpr int ST_PC2 = 68;                      // row stride of a parity plane (67 even / 66 odd columns)
constexpr int ST_LDW = 65;                      // weight row stride

struct StemArgs {
    const float* X; const float* Wt; float* Y;
    int Nb, C, H, W, Ho, Wo;
    int tiles_x, tiles_y;
};

// Round 5: PERSISTENT workgroups (one per CU, looping over tiles).  The first version staged the weights (76 KB out of L2) and the patch for
// every tile and ran load -> compute -> store strictly one after the other: 54 us per tile against 31 us of matrix work (two waves per
// SIMD), 5.6 tiles per CU at batch 24.  Now the weights are staged once per workgroup, and the NEXT tile's patch is fetched into registers
// (33 per thread for six channels) in front of the current tile's MFMA loop - in flight during it - and written to LDS behind it; the
// current tile's output stores are issued after that, so they drain under the next tile's loop.  Same arithmetic, same order: bit-identical.
template <int CP>
__global__ void __launch_bounds__(ST_NT) k_conv7s2_stem(StemArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Wl = smem;                                           // [CP * 49][2][ST_LDW]
    float* patch = smem + CP * 49 * 2 * ST_LDW;                 // [2 CP][ST_PR][2][ST_PC2]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int C = a.C;
    const int ntiles = a.Nb * a.tiles_y * a.tiles_x;
    constexpr int per_c = ST_PR * ST_PW, totp = 2 * CP * per_c, NV = (totp + ST_NT - 1) / ST_NT;
    const __amdgpu_buffer_rsrc_t rsX = fd_make_rsrc(a.X);       // the whole tensor: < 2^29 floats (size guard of the entry point)
    float v[NV];
    // the tile's patch -> registers: element e = tid + u * ST_NT of [2 CP][ST_PR][ST_PW]; out-of-image elements and the padding channel read 0.0
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        int blk = tile;
        const int bx = blk % a.tiles_x; blk /= a.tiles_x;
        const int by = blk % a.tiles_y;
        const int n = blk / a.tiles_y;
        const int iy0 = 2 * by * ST_TR - 3, ix0 = 2 * bx * ST_TC - 3;
        const unsigned nb = (unsigned)n * (unsigned)C * (unsigned)(a.H * a.W);
        int t0 = tid;
        asm volatile("" : "+v"(t0));        // opaque per call: the element -> (channel, row, column) arithmetic is the same for every tile, and hoisted out of
                                            // the tile loop it costs ~130 registers (256 + spills instead of ~130)
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = t0 + u * ST_NT;
            const int c = e / per_c;
            const int r2 = e - c * per_c;
            const int r = r2 / ST_PW, j = r2 - r * ST_PW;
            const int iy = iy0 + r, ix = ix0 + j;
            const bool in = e < totp && c < C && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            v[u] = fd_ldg32(rsX, in ? 4u * (nb + (unsigned)((c * a.H + iy) * a.W + ix)) : FD_OOB);       // out of range reads 0
        }
    };
    // registers -> LDS, de-interleaved by column parity
    auto stash = [&]() __attribute__((always_inline)) {
        int t0 = tid;
        asm volatile("" : "+v"(t0));
#pragma unroll
        for (int u = 0; u < NV; ++u) {
            const int e = t0 + u * ST_NT;
            if (e < totp) {
                const int c = e / per_c;
                const int r2 = e - c * per_c;
                const int r = r2 / ST_PW, j = r2 - r * ST_PW;
                patch[((c * ST_PR + r) * 2 + (j & 1)) * ST_PC2 + (j >> 1)] = v[u];
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) fetch(tile);
    // ---- weights, once per workgroup: flat OIHW read (coalesced), scattered to [k = (cp, t)][parity][m], in batches of 8 independent loads
    {
        const __amdgpu_buffer_rsrc_t rsW = fd_make_rsrc(a.Wt);
        const int total = 64 * C * 49;
        for (int f0 = tid; f0 < total; f0 += 8 * ST_NT) {
            float w8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * ST_NT;
                w8[u] = fd_ldg32(rsW, f < total ? 4u * (unsigned)f : FD_OOB);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int f = f0 + u * ST_NT;
                if (f < total) {
                    const int m = f / (C * 49);
                    const int r = f - m * (C * 49);
                    const int c = r / 49, t = r - c * 49;
                    Wl[(((c >> 1) * 49 + t) * 2 + (c & 1)) * ST_LDW + m] = w8[u];
                }
            }
        }
        if (C & 1) {                                            // the padding channel of the last pair
            for (int f = tid; f < 49 * 64; f += ST_NT) {
                const int t = f >> 6, m = f & 63;
                Wl[(((CP - 1) * 49 + t) * 2 + 1) * ST_LDW + m] = 0.f;
            }
        }
    }
    if (tile < ntiles) stash();
    __syncthreads();

    typedef float f32x16 __attribute__((ext_vector_type(16)));
    const int arow = lane >> 5, l31 = lane & 31;
    const float* pa = Wl + arow * ST_LDW + l31;                                              // + k-step * 2 LDW (+ 32: channels 32..63)
    const float* pb = patch + ((arow * ST_PR + 2 * wave) * 2) * ST_PC2 + l31;               // + tap offset (+ 32: columns 32..63)
    const size_t cs = (size_t)a.Ho * a.Wo;
    for (; tile < ntiles; tile += (int)gridDim.x) {
        const int next = tile + (int)gridDim.x;
        if (next < ntiles) fetch(next);                         // in flight during the MFMA loop
        __builtin_amdgcn_sched_barrier(0);                      // (nothing of the stash's address arithmetic hoisted above the loop: registers)
        f32x16 acc[2][2];                                       // [channel block][column block]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 1
        for (int cp = 0; cp < CP; ++cp) {
            const float* qa = pa + cp * 49 * 2 * ST_LDW;
            const float* qb = pb + cp * 2 * ST_PR * 2 * ST_PC2;
            float a0 = qa[0], a1 = qa[32], b0 = qb[0], b1 = qb[32];
#pragma unroll
            for (int t = 0; t < 49; ++t) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (t + 1 < 49) {
                    const int t1 = t + 1, ky = t1 / 7, kx = t1 - 7 * ky;
                    const int ob = (ky * 2 + (kx & 1)) * ST_PC2 + (kx >> 1);
                    na0 = qa[t1 * 2 * ST_LDW]; na1 = qa[t1 * 2 * ST_LDW + 32];
                    nb0 = qb[ob]; nb1 = qb[ob + 32];
                }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                        // every wave is done with this tile's patch
        if (next < ntiles) stash();
        // ---- epilogue (C/D layout: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)); issued behind the LDS writes of
        //      the next patch: the stores drain under the next tile's loop
        int blk = tile;
        const int bx = blk % a.tiles_x; blk /= a.tiles_x;
        const int by = blk % a.tiles_y;
        const int n = blk / a.tiles_y;
        const int oy = by * ST_TR + wave, ox0 = bx * ST_TC;
        if (oy < a.Ho) {
            float* yn = a.Y + (size_t)n * 64 * a.Ho * a.Wo + (size_t)oy * a.Wo;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int ox = ox0 + 32 * j + l31;
                    if (ox < a.Wo) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * arow;
                            yn[m * cs + ox] = acc[i][j][r];
                        }
                    }
                }
        }
        __syncthreads();                                        // the next patch is in LDS
    }
}

// CUs of the current device (256 on MI355X), asked once
inline int stem_num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

template <int CP>
int stem_go(const StemArgs& a, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)CP * 49 * 2 * ST_LDW + (size_t)2 * CP * ST_PR * 2 * ST_PC2);
    static FdLdsAttrOnce attr_set;
    if (attr_set.needed()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv7s2_stem<CP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set.mark();
    }
    const int ntiles = a.Nb * a.tiles_y * a.tiles_x;
    // one workgroup per CU (145 KB of LDS for six channels), looping over the tiles; with fewer tiles than CUs one tile each
    const int nwg = ntiles < stem_num_cus() ? ntiles : stem_num_cus();
    hipLaunchKernelGGL(k_conv7s2_stem<CP>, dim3((unsigned)nwg), dim3(ST_NT), lds, st, a);
    FD_LAUNCH_CHECK("k_conv7s2_stem");
    return 0;
}

}  // namespace

// The encoder stems: 7x7, stride 2, padding 3 (zeros), 2..6 input channels, 64 output channels, no activation, plain input
bool stem7_fwd_ok(const fd_conv_desc* d) {
    return fd_tun().stem7 != 0 && d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->pad_mode == 0 && d->Cout == 64 &&
           d->Cin >= 1 && d->Cin <= 6 && d->act == 0 && !d->in_norm && d->H >= 8 && d->W >= 8 &&
           (long)d->N * 64 * ((d->H - 1) / 2 + 1) * ((d->W - 1) / 2 + 1) < (1L << 31);
}

int stem7_fwd_launch(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st) {
    if (bias) { fd_set_error("stem7: bias not supported (the ResNet stems have none)"); return -1; }
    StemArgs a;
    a.X = x; a.Wt = w; a.Y = y;
    a.Nb = d->N; a.C = d->Cin; a.H = d->H; a.W = d->W;
    a.Ho = (d->H + 2 * 3 - 7) / 2 + 1; a.Wo = (d->W + 2 * 3 - 7) / 2 + 1;
    a.tiles_x = fd_cdiv(a.Wo, ST_TC); a.tiles_y = fd_cdiv(a.Ho, ST_TR);
    const int cp = (d->Cin + 1) / 2;
    if (cp == 1) return stem_go<1>(a, st);
    if (cp == 2) return stem_go<2>(a, st);
    return stem_go<3>(a, st);
}
