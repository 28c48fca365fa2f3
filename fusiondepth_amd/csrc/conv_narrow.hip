// Weight gradient of the narrow 3x3 stride-1 pad-1 convolutions of the DepthDecoder's full-resolution levels
// (networks/depth_decoder.py:63-96: upconv(0,*) 16/32 -> 16 channels, dispconv(0/1) 16/32 -> 1): Cin in {16, 32}, Cout <= 16;
// and, with two row groups (MG = 2), the 32 -> 32 layers of the Refiner's decoder at the two finest levels
// (networks/refine_decoder.py, 22 -> 32 after channel padding), which the generic gather kernel took 125 / 510 us for.
//
// As a GEMM this is M = Cout <= 16 rows x J = 9*Cin columns x K = N*H*W pixels: 0.1 GFLOP per megapixel of work on 190 MB
// of operands - memory-bound.  The generic gather kernel pads M and J up to 32x128 MFMA tiles and gathers every tap
// separately (reflect logic per element): 450-640 us per layer on the decoder's serial critical path.  Here
//   * a workgroup stages a 4 x 64 pixel tile of dY and the matching (4+2) x (64+2) halo patch of X in LDS ONCE and forms
//     the nine tap-shifted operands by reading the patch at shifted addresses (no 9x operand expansion);
//   * v_mfma_f32_16x16x4_f32 fits the problem exactly: 16 (Cout) x 16 (channels of one group) outputs per tap, K = 4
//     consecutive pixels; one wave per tile row, 9 x (Cin/16) accumulators of 4 VGPRs per wave held across all tiles;
//   * LDS strides are = 4 (mod 64) words so the 64 lanes of an operand read (16 rows x 4 pixels) hit 64 distinct banks;
//   * persistent workgroups (<= 512) loop over the tiles; per-workgroup partial sums go to slabs that a second kernel
//     adds in a fixed order (deterministic, no float atomics).
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_narrow.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TW = 64, TH = 4;                 // pixel tile
constexpr int PR = TH + 2, PCW = TW + 2;       // patch rows / columns
constexpr int XRS = 68;                        // patch row stride (floats)
constexpr int XCS = 452;                       // patch channel stride: >= PR * XRS and = 4 (mod 64)
constexpr int YS = 260;                        // dY row stride: >= TH * TW and = 4 (mod 64)
static_assert(XCS >= PR * XRS && XCS % 64 == 4 && YS >= TH * TW && YS % 64 == 4, "LDS strides");

__device__ __forceinline__ int refl_clamp_idx(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);   // far-out halo positions of partial tiles: any in-range value (their dY is 0)
}

template <int CG, int MG = 1>   // input-channel groups of 16, output-row groups of 16
__global__ void __launch_bounds__(256) k_wgrad_narrow(NarrowWgradArgs g) {
    constexpr int C = 16 * CG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                          // [C][XCS]
    float* sY = smem + C * XCS;                // [16 * MG][YS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsX = fd_make_rsrc(g.X), rsY = fd_make_rsrc(g.dY);
    const int tiles_x = (g.W + TW - 1) / TW, tiles_y = (g.H + TH - 1) / TH;
    const int tiles_per_img = tiles_x * tiles_y;
    const int ntiles = g.N * tiles_per_img;
    const unsigned hw = (unsigned)(g.H * g.W);

    f32x4 acc[9][CG * MG];                      // [tap][cg * MG + mg]
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < CG * MG; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int n = tile / tiles_per_img, r = tile - n * tiles_per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();                        // previous tile's readers are done
        // ---- X halo patch (rows y0-1 .. y0+TH, columns x0-1 .. x0+TW) and the dY tile (rows >= M and pixels outside the
        //      image read 0): the loads of a 16-channel group are all issued before its first LDS store (latencies overlap)
        constexpr int NXI = (16 * PR * PCW + 255) / 256, NYI = MG * 16 * TH * TW / 256;   // X: per 16-channel group (bounds the registers)
        float vy[NYI];
#pragma unroll
        for (int it = 0; it < NYI; ++it) {
            const int i = tid + it * 256;
            const int m = i / (TH * TW), p = i - m * (TH * TW);
            const int py = p / TW, px = p - py * TW;
            const bool ok = (m < g.M) & (y0 + py < g.H) & (x0 + px < g.W);
            vy[it] = fd_ldg32(rsY, ok ? 4u * (((unsigned)(n * g.M + m)) * hw + (unsigned)((y0 + py) * g.W + x0 + px)) : FD_OOB);
        }
#pragma unroll
        for (int cg = 0; cg < CG; ++cg) {
            float vx[NXI];
#pragma unroll
            for (int it = 0; it < NXI; ++it) {
                const int i = tid + it * 256;
                const int c = i / (PR * PCW), q = i - c * (PR * PCW);
                const int pr = q / PCW, pc = q - pr * PCW;
                const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
                const unsigned plane = ((unsigned)(n * C + cg * 16 + c)) * hw;
                unsigned off;
                if (g.pad_mode == 1) {
                    off = 4u * (plane + (unsigned)(refl_clamp_idx(yy, g.H) * g.W + refl_clamp_idx(xx, g.W)));
                } else {
                    const bool in = ((unsigned)yy < (unsigned)g.H) & ((unsigned)xx < (unsigned)g.W);
                    off = in ? 4u * (plane + (unsigned)(yy * g.W + xx)) : FD_OOB;
                }
                vx[it] = fd_ldg32(rsX, i < 16 * PR * PCW ? off : FD_OOB);
            }
            if (cg == 0) {
#pragma unroll
                for (int it = 0; it < NYI; ++it) {
                    const int i = tid + it * 256;
                    const int m = i / (TH * TW), p = i - m * (TH * TW);
                    sY[m * YS + p] = vy[it];
                }
            }
#pragma unroll
            for (int it = 0; it < NXI; ++it) {
                const int i = tid + it * 256;
                const int c = i / (PR * PCW), q = i - c * (PR * PCW);
                const int pr = q / PCW, pc = q - pr * PCW;
                if (i < 16 * PR * PCW) sX[(cg * 16 + c) * XCS + pr * XRS + pc] = vx[it];
            }
        }
        __syncthreads();
        // ---- wave `wave` owns tile row `wave`: 16 k-steps of 4 pixels
        const float* pa = sY + li * YS + wave * TW + lk;
        const float* pb = sX + li * XCS + wave * XRS + lk;
#pragma unroll 2
        for (int ks = 0; ks < TW / 4; ++ks) {
            float a[MG];
#pragma unroll
            for (int mg = 0; mg < MG; ++mg) a[mg] = pa[mg * 16 * YS + 4 * ks];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        const float b = pb[c * 16 * XCS + dy * XRS + 4 * ks + dx];
#pragma unroll
                        for (int mg = 0; mg < MG; ++mg)
                            acc[dy * 3 + dx][c * MG + mg] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mg], b, acc[dy * 3 + dx][c * MG + mg], 0, 0, 0);
                    }
        }
    }
    // ---- sum the four waves (fixed order) and write this workgroup's partial slab [m][c][tap]
    float* slab = g.slabs + (size_t)blockIdx.x * g.M * C * 9;
    float* red = smem;                          // [4 waves][CG * MG][64 lanes][4 regs]
    constexpr int NA = CG * MG;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NA; ++c)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) red[((wave * NA + c) * 64 + lane) * 4 + rg] = acc[t][c][rg];
        __syncthreads();
        for (int i = tid; i < NA * 256; i += 256) {
            const int c = i >> 8, q = i & 255, ln = q >> 2, rg = q & 3;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[((w * NA + c) * 64 + ln) * 4 + rg];
            // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
            const int m = (c % MG) * 16 + 4 * (ln >> 4) + rg, ch = (c / MG) * 16 + (ln & 15);
            if (m < g.M) slab[(m * C + ch) * 9 + t] = s;
        }
    }
}

// out[i] (+)= sum_z slabs[z][i], z-parallel: 8 thread groups sum contiguous z ranges, combined in group order.
__global__ void __launch_bounds__(256) k_reduce_slabs_z(const float* __restrict__ slabs, float* __restrict__ out, int n, int nslab,
                                                        int accumulate) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + e;
    const int per = (nslab + 7) / 8;
    const int z0 = zg * per, z1 = z0 + per < nslab ? z0 + per : nslab;
    float s = 0.f;
    if (i < n) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {                    // eight loads in flight, added in slab order (same sum as the plain loop)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(z + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < z1; ++z) s += slabs[(size_t)z * n + i];
    }
    part[zg][e] = s;
    __syncthreads();
    if (zg == 0 && i < n) {
        float t = accumulate ? out[i] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += part[k][e];
        out[i] = t;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 7x7 stride-2 pad-3 ResNet stems (networks/resnet_encoder.py:92, Cin = 2 / 3 / 4 / 6 after the
// input-channel variants of :53-76): M = 64, J = 49*Cin, K = N*Ho*Wo.  Same structure as above: a 4 x 32 output tile of dY
// (64 rows) and the (2*4+5) x (2*32+5) input patch per channel in LDS, zero padding resolved while staging.  The MFMA
// N axis is a block of taps: column n = (dy & 1, dx) of a row pair -> 16 columns per (row pair, channel), 49 of 64 used.
// Wave w owns dY rows 16w .. 16w+15 and walks all 128 pixels of the tile: 4 row pairs x Cin accumulators.
constexpr int STW = 32, STH = 4;                       // output-pixel tile
constexpr int SPR = 2 * STH + 5, SPC = 2 * STW + 5;     // input patch rows / columns (13 x 69)
constexpr int SXRS = 70;                               // patch row stride
constexpr int SXCS = SPR * SXRS + 2;                   // patch channel stride (912)
constexpr int SYS = 132;                               // dY row stride: >= 128 and = 4 (mod 64)
static_assert(SYS >= STH * STW && SYS % 64 == 4, "stem LDS strides");

template <int C>
__global__ void __launch_bounds__(256) k_wgrad_stem(NarrowWgradArgs g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sX = smem;                          // [C][SXCS]
    float* sY = smem + C * SXCS;               // [64][SYS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsXs = fd_make_rsrc(g.X - (3 * g.W + 3)), rsY = fd_make_rsrc(g.dY);
    const int Ho = (g.H + 6 - 7) / 2 + 1, Wo = (g.W + 6 - 7) / 2 + 1;
    const int tiles_x = (Wo + STW - 1) / STW, tiles_y = (Ho + STH - 1) / STH;
    const int tiles_per_img = tiles_x * tiles_y;
    const int ntiles = g.N * tiles_per_img;
    const unsigned hw = (unsigned)(g.H * g.W), howo = (unsigned)(Ho * Wo);

    f32x4 acc[4][C];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int c = 0; c < C; ++c) acc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B operand of column n = li: tap row parity li >> 3, tap column li & 7 (column 7 and row 7 are padding, never stored)
    const int boff = (li >> 3) * SXRS + (li & 7) + 2 * lk;

    // Software pipeline over the workgroup's tiles: the global loads of tile t + 1 (patch + all of dY: NXI + 32 registers) are issued
    // in front of the MFMA loop of tile t and written to LDS behind it.  (Round 3 loaded, stored and computed one tile after the
    // other and left the overlap to the second workgroup of the CU: 45 % of the matrix floor, the 2- and 3-channel stems bound by
    // the load phase.)
    // Thread -> element maps with ONE vector offset per operand and tile: dY element (row 2 it + hi, pixel tid & 127), patch element
    // (channel c, row 2 k + hi, column tid & 127 < 69); everything that varies with it / (c, k) is a wave-uniform scalar offset of the
    // buffer load resp. an immediate of the LDS store.  (With the flat index tid + 256 it decomposed per load, hipcc hoisted 53
    // offsets + masks out of the tile loop: 365 registers for six channels.)
    constexpr int NYI = 64 * STH * STW / 256, NXK = (SPR + 1) / 2;
    float vx[C][NXK], vy[NYI];
    const int hi = tid >> 7, q7 = tid & 127;
    const int qy = q7 >> 5, qx = q7 & 31;
    auto fetch = [&](int tile) __attribute__((always_inline)) {
        const bool live = tile < ntiles;
        const int tl = live ? tile : 0;
        const int n = tl / tiles_per_img, r = tl - n * tiles_per_img;
        const int ty = r / tiles_x, tx = r - ty * tiles_x;
        const int y0 = ty * STH, x0 = tx * STW;                 // output coordinates
        const int iy0 = 2 * y0 - 3, ix0 = 2 * x0 - 3;           // input coordinates of the patch origin
        const int xx = ix0 + q7;
        const bool colok = live & (q7 < SPC) & ((unsigned)xx < (unsigned)g.W);
        // rsXs starts 3 rows + 3 pixels in front of X, so the offset of (row iy0 + hi, column xx) is never negative (a wrapped
        // offset would fail the range check for the rows 2 k below it that ARE inside the image)
        const unsigned xb = 4u * ((unsigned)(n * C) * hw + (unsigned)((iy0 + hi + 3) * g.W + xx + 3));
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int k = 0; k < NXK; ++k) {
                const int yy = iy0 + hi + 2 * k;
                const bool in = colok & ((unsigned)yy < (unsigned)g.H) & (2 * k + hi < SPR);
                vx[c][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsXs, (int)(in ? xb : FD_OOB), (int)(4u * ((unsigned)c * hw + (unsigned)(2 * k * g.W))), 0));
            }
        const bool ok = live & (y0 + qy < Ho) & (x0 + qx < Wo);
        const unsigned yb = 4u * (((unsigned)(n * g.M + hi)) * howo + (unsigned)((y0 + qy) * Wo + x0 + qx));
#pragma unroll
        for (int it = 0; it < NYI; ++it)
            vy[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsY, (int)((ok & (2 * it + hi < g.M)) ? yb : FD_OOB), (int)(4u * 2u * (unsigned)it * howo), 0));
    };
    auto stage = [&]() __attribute__((always_inline)) {
        if (q7 < SPC) {
#pragma unroll
            for (int c = 0; c < C; ++c)
#pragma unroll
                for (int k = 0; k < NXK; ++k)
                    if (2 * k + hi < SPR) sX[c * SXCS + (2 * k + hi) * SXRS + q7] = vx[c][k];
        }
#pragma unroll
        for (int it = 0; it < NYI; ++it) sY[(2 * it + hi) * SYS + q7] = vy[it];
    };
    fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        __syncthreads();                                        // the previous tile's operand reads are done
        stage();
        __syncthreads();
        fetch(tile + gridDim.x);                                // in flight during the MFMA loop below
        const float* pa = sY + (wave * 16 + li) * SYS + lk;
#pragma unroll 1
        for (int py = 0; py < STH; ++py) {
#pragma unroll 2
            for (int ks = 0; ks < STW / 4; ++ks) {
                const float a = pa[py * STW + 4 * ks];
                const float* pb = sX + (2 * py) * SXRS + 8 * ks + boff;     // input row 2*py + dy, column 2*(4ks + k) + dx
#pragma unroll
                for (int dp = 0; dp < 4; ++dp)
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const float b = pb[c * SXCS + 2 * dp * SXRS];
                        acc[dp][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[dp][c], 0, 0, 0);
                    }
            }
        }
    }
    // ---- this workgroup's partial slab [m][c][7][7]; wave w holds rows 16w..16w+15 (no cross-wave sum needed)
    float* slab = g.slabs + (size_t)blockIdx.x * g.M * C * 49;
#pragma unroll
    for (int dp = 0; dp < 4; ++dp)
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int m = wave * 16 + 4 * lk + rg;             // C/D layout: row = 4*(lane>>4)+reg, col = lane&15
                const int dy = 2 * dp + (li >> 3), dx = li & 7;
                if (m < g.M && dy < 7 && dx < 7) slab[(m * C + c) * 49 + dy * 7 + dx] = acc[dp][c][rg];
            }
}

}  // namespace

bool stem_wgrad_ok(const fd_conv_desc* d) {
    return d->KH == 7 && d->KW == 7 && d->stride == 2 && d->pad == 3 && d->pad_mode == 0 && !d->in_norm && d->Cout <= 64 &&
           (d->Cin == 2 || d->Cin == 3 || d->Cin == 4 || d->Cin == 6) && (double)d->N * d->Cout * d->H * d->W < 2147483648.0;
}
static int stem_blocks(const fd_conv_desc* d) {
    const int Ho = (d->H - 1) / 2 + 1, Wo = (d->W - 1) / 2 + 1;
    const long tiles = (long)d->N * ((Ho + STH - 1) / STH) * ((Wo + STW - 1) / STW);
    return (int)(tiles < 512 ? tiles : 512);
}
long stem_wgrad_ws_floats(const fd_conv_desc* d) { return (long)stem_blocks(d) * d->Cout * d->Cin * 49; }

int stem_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate, hipStream_t st) {
    NarrowWgradArgs a;
    a.X = x; a.dY = gy; a.slabs = ws; a.N = d->N; a.M = d->Cout; a.H = d->H; a.W = d->W; a.pad_mode = 0;
    const int blocks = stem_blocks(d);
    const size_t lds = sizeof(float) * ((size_t)d->Cin * SXCS + 64 * SYS);
    switch (d->Cin) {
        case 2: hipLaunchKernelGGL(k_wgrad_stem<2>, dim3(blocks), dim3(256), lds, st, a); break;
        case 3: hipLaunchKernelGGL(k_wgrad_stem<3>, dim3(blocks), dim3(256), lds, st, a); break;
        case 4: hipLaunchKernelGGL(k_wgrad_stem<4>, dim3(blocks), dim3(256), lds, st, a); break;
        default: hipLaunchKernelGGL(k_wgrad_stem<6>, dim3(blocks), dim3(256), lds, st, a); break;
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_wgrad_stem launch failed: %s", hipGetErrorString(e)); return (int)e; }
    const int n = d->Cout * d->Cin * 49;
    hipLaunchKernelGGL(k_reduce_slabs_z, dim3((n + 31) / 32), dim3(256), 0, st, ws, gw, n, blocks, accumulate);
    e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_reduce_slabs_z launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}

bool narrow_wgrad_ok(const fd_conv_desc* d) {
    return d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && !d->in_norm && (d->Cin == 16 || d->Cin == 32) &&
           (d->Cout <= 16 || (d->Cout <= 32 && d->Cin == 32)) && (double)d->N * d->Cin * d->H * d->W * 4.0 < 2147483648.0;
}

int narrow_wgrad_blocks(const fd_conv_desc* d) {
    const long tiles = (long)d->N * ((d->H + TH - 1) / TH) * ((d->W + TW - 1) / TW);
    return (int)(tiles < 512 ? tiles : 512);
}

long narrow_wgrad_ws_floats(const fd_conv_desc* d) { return (long)narrow_wgrad_blocks(d) * d->Cout * d->Cin * 9; }

int narrow_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate,
                        hipStream_t st) {
    NarrowWgradArgs a;
    a.X = x; a.dY = gy; a.slabs = ws; a.N = d->N; a.M = d->Cout; a.H = d->H; a.W = d->W; a.pad_mode = d->pad_mode;
    const int blocks = narrow_wgrad_blocks(d);
    const int cg = d->Cin / 16, mg = (d->Cout + 15) / 16;
    const size_t lds = sizeof(float) * ((size_t)d->Cin * XCS + 16 * mg * YS);
    if (cg == 1) {
        hipLaunchKernelGGL(k_wgrad_narrow<1>, dim3(blocks), dim3(256), lds, st, a);
    } else if (mg == 2) {                       // the Refiner decoder's 32 -> 32 layers (refine_decoder: upconv(1,*) / (0,*) after padding)
        static FdLdsAttrOnce attr2;
        if (attr2.needed()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_narrow<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      160 * 1024);
            attr2.mark();
        }
        hipLaunchKernelGGL((k_wgrad_narrow<2, 2>), dim3(blocks), dim3(256), lds, st, a);
    } else {
        static FdLdsAttrOnce attr;
        if (attr.needed()) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_wgrad_narrow<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr.mark();
        }
        hipLaunchKernelGGL(k_wgrad_narrow<2>, dim3(blocks), dim3(256), lds, st, a);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_wgrad_narrow launch failed: %s", hipGetErrorString(e)); return (int)e; }
    const int n = d->Cout * d->Cin * 9;
    hipLaunchKernelGGL(k_reduce_slabs_z, dim3((n + 31) / 32), dim3(256), 0, st, ws, gw, n, blocks, accumulate);
    e = hipGetLastError();
    if (e != hipSuccess) { fd_set_error("k_reduce_slabs_z launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
