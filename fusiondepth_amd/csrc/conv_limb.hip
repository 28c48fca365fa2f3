// Split-precision ("3 x bf16 limbs, six products, fp32 accumulate") kernels for the convolutions that have no Winograd form:
//   * 1x1 stride-1 - the bottleneck blocks of a ResNet-50 (torchvision Bottleneck conv1 / conv3 / downsample: networks/resnet_encoder.py:62-74),
//     the PoseDecoder's squeeze layer (networks/pose_decoder.py:20): k_gemm_limb (forward, data gradient), k_wgrad_limb;
//   * stride 2 - layerN.0.conv1 (BasicBlock) / conv2 (Bottleneck) 3x3 and the large 1x1 downsample layers of every ResNet: k_conv_limb
//     (forward), k_conv_limb_grp (the four output-parity classes of the data gradient in one launch), k_wgrad_limb_s2.
//
// Why: such a convolution is a plain (implicit) GEMM; on v_mfma_f32_32x32x2_f32 it runs at the fp32 matrix rate (157 TFLOP/s peak, 58 - 100
// measured on these shapes).  v_mfma_f32_32x32x16_bf16 is 16x faster per instruction; with every fp32 operand split into three bf16 limbs
// (conv_limb.h) six of them reproduce the fp32 product: 2.7x the fp32 matrix peak at fp32 accuracy.  Inside the training step, where four
// streams share the matrix pipes, the matrix cycles saved count even where a kernel's stand-alone time does not move
// (profiles/round6_experiments.md section 3).
//
// Shape of the kernels (CDNA4: 64-lane waves, 4 SIMDs per CU, 160 KB LDS):
//   * forward / data gradient  Y[n][m][p] = sum_k A[m][k] X[n][k][p]:  A = the weights, PRE-SPLIT once per optimiser step into the
//     LDS image of a K-chunk (conv_limb.h; K = channel for the 1x1 layers, (tap, channel) for k_conv_limb) -> plain 16-byte copies
//     global -> LDS, no arithmetic; X is read as fp32 (coalesced along the pixels): lane l of a wave loads 8 consecutive channels of
//     its pixel, splits them in registers (52 vector instructions) and HAS its three MFMA B fragments - no LDS round trip for X;
//   * weight gradient  dW[m][c] = sum_{n,p} dY[n][m][p] X[n][c][p]:  both operands are K(= pixel)-contiguous fp32 rows; a thread
//     owns 8 pixels of a row, same split, 16-byte LDS stores.  Split over K into slabs, reduced by the existing fixed-order finish
//     kernel (deterministic, no atomics); the stride-2 form takes one tap per workgroup (see k_wgrad_limb_s2);
//   * per K-step of 16 a wave reads 3 fragments per 32 x 32 block and operand and issues 6 MFMAs per block (192 matrix-pipe cycles).
#include "../../include/fdhip.h"
#include "fd_common.h"
#include "conv_fast.h"
#include "conv_limb.h"
#include <type_traits>

namespace {
using namespace fdlimb;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct LimbGemmArgs {
    const void* A3;        // pre-split weights (conv_limb.h layout), M rows, K columns
    const float* X;        // [Nb][K][HW]
    float* Y;              // [Nb][M][HW]
    const float* bias;     // [M] or null
    const float* add;      // laid out like Y or null: Y = act(. + bias) + add
    float* slabs;          // split-K: [splits][Nb][M][HW]
    long slab_stride;
    int M, K, Nb, HW;
    int act;               // 0 none, 1 ReLU
    int ntm, ntn;          // channel tiles, pixel tiles
};

__device__ __forceinline__ uint4 ldg_u128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ bf16x8 as_frag(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// the six limb products of one 32 x 32 x 16 block step; (l, h) and (h, l) first, the large terms last
#define FD_LIMB_MFMA6(ACC, AF, BF)                                                                          \
    do {                                                                                                    \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[2]), as_frag(BF[0]), ACC, 0, 0, 0);          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[0]), as_frag(BF[2]), ACC, 0, 0, 0);          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[1]), as_frag(BF[1]), ACC, 0, 0, 0);          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[1]), as_frag(BF[0]), ACC, 0, 0, 0);          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[0]), as_frag(BF[1]), ACC, 0, 0, 0);          \
        ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag(AF[0]), as_frag(BF[0]), ACC, 0, 0, 0);          \
    } while (0)

// ------------------------------------------------------------------------------------------------ forward / data gradient
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Workgroup = 4 waves, tile (32 MB) channels x 128 pixels; a wave owns ALL the tile's channels x 32 pixels (MB blocks of 32 x 32).
//   * X never touches LDS: lane l of a wave loads the 8 channels 16 c + 8 (l / 32) .. + 7 of pixel l % 32 (8 dword loads, two
//     128-byte lines per wave-instruction), splits them in registers and HAS its three MFMA B fragments - no store, no barrier, no
//     read-back, and a wave's X loads depend on nobody else.  They run D chunks ahead in a statically indexed register ring.
//   * the weights (pre-split, L2-resident) are shared by the four waves: registers -> LDS (double-buffered, one barrier per chunk),
//     fetched D chunks ahead as well; a wave reads 3 MB fragments per chunk for 6 MB MFMAs.
//   * the chunk loop is unrolled D times with no branch inside (K % (16 D) == 0: the launcher guarantees it), the prefetch past the
//     end re-reads the last chunk instead of branching; chunk and channel offsets ride in the loads' SCALAR offset, so the loop has
//     no vector address arithmetic at all.
// History (profiles/round6_limb_ab_v*.log, round6_limb_v4_*_pmc.md): with X staged through LDS as well (a thread splits, stores, all
// waves read back behind a barrier) the matrix pipes were busy 27 % of the kernel and the waves spent 45 % of their cycles in issue
// stalls and 32 % parked on barriers / counters, whatever the prefetch depth - 12 MFMAs per wave and barrier cannot hide two LDS
// round trips.
// grid.x = 8 * ceil(ntn / 8) * ntm: workgroup id -> XCD = id % 8 (the hardware's round-robin), and inside an XCD the tiles that
// share a pixel tile (all ntm channel tiles) follow each other, so X comes from HBM once and from that XCD's L2 afterwards.
template <int MB, int D>
__global__ void __launch_bounds__(256, 2) k_gemm_limb(LimbGemmArgs g) {
    constexpr int NT = 256;
    constexpr int BM = 32 * MB, BN = 128;
    constexpr int A_BYTES = 96 * BM;                                  // one chunk: [3 limbs][2 halves][BM rows][16 B]
    constexpr int A_PIECES = 6 * BM, NA = (A_PIECES + NT - 1) / NT;
    static_assert(D % 2 == 0, "the LDS buffer parity of a chunk must be a compile-time constant inside the unrolled body");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [2][A_BYTES]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = (int)blockIdx.x >> 3;
    const int tm = slot % g.ntm, tn = (slot / g.ntm) * 8 + ((int)blockIdx.x & 7);
    if (tn >= g.ntn) return;
    const int m0 = tm * BM;
    const long p0 = (long)tn * BN, Np = (long)g.Nb * g.HW;
    const int nch = g.K >> 4;
    const int nsplit = (int)gridDim.y, zs = (int)blockIdx.y;
    const int per = nch / nsplit;                                     // a multiple of D (launcher)
    const int c_lo = zs * per, c_hi = c_lo + per;

    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.A3), rsX = fd_make_rsrc(g.X);
    // ---- A pieces of this thread: q = tid + NT i -> (limb-half lh = q / BM, row = q % BM)
    unsigned a_base[NA];
    const unsigned a_step = 96u * (unsigned)g.M;             // bytes per K-chunk: 6 (limb, half) x M pieces x 16
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + NT * i;
        const int lh = q / BM, row = q - lh * BM;
        int m = m0 + row;
        m = m < g.M ? m : g.M - 1;
        a_base[i] = (A_PIECES % NT != 0 && q >= A_PIECES) ? FD_OOB : 16u * ((unsigned)lh * (unsigned)g.M + (unsigned)m);
    }
    // ---- this lane's X column: pixel wave * 32 + lane % 32, channels 8 (lane / 32) .. + 7 of every chunk
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned hw4 = 4u * (unsigned)g.HW;
    unsigned b_base;
    {
        const long pg = p0 + wave * 32 + l31;
        const bool ok = pg < Np;
        const long pp = ok ? pg : 0;
        const int n = (int)(pp / g.HW);
        const int pix = (int)(pp - (long)n * g.HW);
        b_base = ok ? 4u * (unsigned)(((long)n * g.K + 8 * half) * g.HW + pix) : FD_OOB;
    }
    uint4 ra[D][NA];
    float rb[D][8];
    auto load_ab = [&](auto slot_tag, int c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const int ce = c < c_hi ? c : c_hi - 1;                // past the end: the last chunk again (never used)
        const unsigned sa = (unsigned)ce * a_step, sb = (unsigned)ce * 16u * hw4;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[S][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)a_base[i], (int)sa, 0));
#pragma unroll
        for (int e = 0; e < 8; ++e)
            rb[S][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (int)b_base, (int)(sb + (unsigned)e * hw4), 0));
    };
    auto store_a = [&](auto slot_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        unsigned char* base = smem + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + NT * i;
            if (A_PIECES % NT == 0 || q < A_PIECES) *reinterpret_cast<uint4*>(base + 16 * q) = ra[S][i];
        }
    };

    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    auto mfma_chunk = [&](auto slot_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        uint4 bf[3];
        split8(rb[S], bf[0], bf[1], bf[2]);
        const unsigned char* sa = smem + buf * A_BYTES + 16 * (half * BM + l31);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            uint4 af[3];
#pragma unroll
            for (int L = 0; L < 3; ++L) af[L] = *reinterpret_cast<const uint4*>(sa + 32 * BM * L + 512 * i);
            FD_LIMB_MFMA6(acc[i], af, bf);
        }
    };
    if (c_lo < c_hi) {
        static_for<D>([&](auto j) __attribute__((always_inline)) { load_ab(j, c_lo + decltype(j)::value); });
        store_a(std::integral_constant<int, 0>{}, 0);
        __syncthreads();
        for (int c = c_lo; c < c_hi; c += D) {
            static_for<D>([&](auto j) __attribute__((always_inline)) {
                constexpr int J = decltype(j)::value;
                mfma_chunk(j, J & 1);                          // consumes rb[J] and LDS buffer J & 1
                store_a(std::integral_constant<int, (J + 1) % D>{}, (J & 1) ^ 1);       // the weights of chunk c + J + 1
                load_ab(j, c + J + D);                         // slot J is free now: chunk c + J + D
                __syncthreads();
            });
        }
    }

    // ---- epilogue (C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    const bool final_pass = nsplit == 1;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(final_pass ? g.Y : g.slabs + (size_t)zs * g.slab_stride);
    const __amdgpu_buffer_rsrc_t rsAdd = fd_make_rsrc(g.add ? g.add : g.Y);
    const bool has_add = final_pass && g.add, has_bias = final_pass && g.bias, relu = final_pass && g.act == 1;
    const unsigned cs4 = 4u * (unsigned)g.HW;
    const long p = p0 + wave * 32 + l31;
    unsigned pix = FD_OOB;
    if (p < Np) {
        const int n = (int)(p / g.HW);
        pix = 4u * (unsigned)((long)n * g.M * g.HW + (p - (long)n * g.HW));
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {                  // four rows at a time: offsets, residual loads in flight, stores
            unsigned off[4];
            float addv[4], bv[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = m0 + i * 32 + r4 + 8 * rq + 4 * half;
                off[r4] = m < g.M ? pix + (unsigned)m * cs4 : FD_OOB;
                bv[r4] = has_bias ? g.bias[m < g.M ? m : g.M - 1] : 0.f;
                addv[r4] = has_add ? fd_ldg32(rsAdd, off[r4]) : 0.f;
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float v = acc[i][rq * 4 + r4] + bv[r4];
                v = relu ? (v > 0.f ? v : 0.f) : v;
                v += addv[r4];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, (int)off[r4], 0, 0);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ implicit-GEMM convolution
// k_gemm_limb with the GEMM-K index running over (tap, channel) - the direct convolutions Winograd cannot take: the 3x3 and 1x1
// STRIDE-2 layers of a ResNet (layer2/3/4.0.conv1 + downsample: networks/resnet_encoder.py, torchvision BasicBlock / Bottleneck) forward,
// and the four output-parity classes of their data gradients in one grouped launch.  Problem description = conv_fast.h's
// FastGemmArgs (g.A = the pre-split weights A3 of the matrix [M][K = (tap, channel)]), zero padding only.
//   * a K-chunk = 16 channels of ONE tap; the lane's pixel offset of a tap is one per-lane value (border test included: a tap that
//     falls outside the image gets the out-of-range offset and reads zeros), the channel part rides in the loads' scalar offset;
//     tap / chunk counters are wave-uniform scalars advanced per load (no division, no divergent branch in the loop);
//   * everything else - X straight from registers into MFMA fragments, weights through double-buffered LDS, branch-free loop unrolled D
//     times, XCD-aware 1-D grid - is k_gemm_limb's; the epilogue takes the affine output map of the parity classes (osy / osx = 2).
template <int MB, int D>
__device__ __forceinline__ void conv_limb_body(const FastGemmArgs& g, unsigned char* smem, int tm, int tn, int zs, int nsplit) {
    constexpr int NT = 256;
    constexpr int BM = 32 * MB, BN = 128;
    constexpr int A_BYTES = 96 * BM;
    constexpr int A_PIECES = 6 * BM, NA = (A_PIECES + NT - 1) / NT;
    static_assert(D % 2 == 0, "LDS buffer parity");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = tm * BM;
    const int plane = g.NY * g.NX;
    const long p0 = (long)tn * BN, Np = (long)g.Nb * plane;
    const int cpt = g.C >> 4;                                         // chunks per tap
    const int nch = g.T * cpt;
    const int per = nch / nsplit;                                     // a multiple of D (launcher)
    const int c_lo = zs * per, c_hi = c_lo + per;

    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.A), rsX = fd_make_rsrc(g.X);
    unsigned a_base[NA];
    const unsigned a_step = 96u * (unsigned)g.M;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int q = tid + NT * i;
        const int lh = q / BM, row = q - lh * BM;
        int m = m0 + row;
        m = m < g.M ? m : g.M - 1;
        a_base[i] = (A_PIECES % NT != 0 && q >= A_PIECES) ? FD_OOB : 16u * ((unsigned)lh * (unsigned)g.M + (unsigned)m);
    }
    const int half = lane >> 5, l31 = lane & 31;
    const unsigned chw = (unsigned)(g.Hi * g.Wi), hw4 = 4u * chw;
    int ry0, cx0;
    unsigned nb8;
    bool pvalid;
    {
        const long pg = p0 + wave * 32 + l31;
        pvalid = pg < Np;
        const long pp = pvalid ? pg : 0;
        const int n = (int)(pp / plane);
        const int rem = (int)(pp - (long)n * plane);
        const int y = rem / g.NX, x = rem - y * g.NX;
        ry0 = y * g.sy + g.oy; cx0 = x * g.sx + g.ox;
        nb8 = ((unsigned)n * (unsigned)g.C + 8u * (unsigned)half) * chw;
    }
    // (tap row, tap column, channel chunk of the tap) of the next chunk to load: wave-uniform
    int pc_ta, pc_tb, pc_cc;
    { const int t = c_lo / cpt; pc_cc = c_lo - t * cpt; pc_ta = t / g.TB; pc_tb = t - pc_ta * g.TB; }

    uint4 ra[D][NA];
    float rb[D][8];
    auto load_ab = [&](auto slot_tag, int c) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const bool live = c < c_hi;
        const int ce = live ? c : c_hi - 1;                    // past the end: the last chunk's weights again, zeros for X (never used)
        const unsigned sa = (unsigned)ce * a_step;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            ra[S][i] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsA, (int)a_base[i], (int)sa, 0));
        const int r = ry0 + pc_ta * g.da, cc = cx0 + pc_tb * g.db;
        const bool inb = ((unsigned)r < (unsigned)g.Hi) & ((unsigned)cc < (unsigned)g.Wi);
        const unsigned boff = (pvalid & inb & live) ? 4u * (nb8 + (unsigned)(r * g.Wi + cc)) : FD_OOB;
        const unsigned sb = (unsigned)pc_cc * 16u * hw4;
#pragma unroll
        for (int e = 0; e < 8; ++e)
            rb[S][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsX, (int)boff, (int)(sb + (unsigned)e * hw4), 0));
        ++pc_cc;
        const bool w1 = pc_cc == cpt;
        pc_cc = w1 ? 0 : pc_cc;
        pc_tb += w1 ? 1 : 0;
        const bool w2 = pc_tb == g.TB;
        pc_tb = w2 ? 0 : pc_tb;
        pc_ta += w2 ? 1 : 0;
    };
    auto store_a = [&](auto slot_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        unsigned char* base = smem + buf * A_BYTES;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int q = tid + NT * i;
            if (A_PIECES % NT == 0 || q < A_PIECES) *reinterpret_cast<uint4*>(base + 16 * q) = ra[S][i];
        }
    };
    f32x16 acc[MB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    auto mfma_chunk = [&](auto slot_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        uint4 bf[3];
        split8(rb[S], bf[0], bf[1], bf[2]);
        const unsigned char* sa = smem + buf * A_BYTES + 16 * (half * BM + l31);
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            uint4 af[3];
#pragma unroll
            for (int L = 0; L < 3; ++L) af[L] = *reinterpret_cast<const uint4*>(sa + 32 * BM * L + 512 * i);
            FD_LIMB_MFMA6(acc[i], af, bf);
        }
    };
    if (c_lo < c_hi) {
        static_for<D>([&](auto j) __attribute__((always_inline)) { load_ab(j, c_lo + decltype(j)::value); });
        store_a(std::integral_constant<int, 0>{}, 0);
        __syncthreads();
        for (int c = c_lo; c < c_hi; c += D) {
            static_for<D>([&](auto j) __attribute__((always_inline)) {
                constexpr int J = decltype(j)::value;
                mfma_chunk(j, J & 1);
                store_a(std::integral_constant<int, (J + 1) % D>{}, (J & 1) ^ 1);
                load_ab(j, c + J + D);
                __syncthreads();
            });
        }
    }
    // ---- epilogue: affine output map (conv_fast.hip), C/D layout of the 32x32 MFMA
    const bool final_pass = nsplit == 1;
    const __amdgpu_buffer_rsrc_t rsY = fd_make_rsrc(final_pass ? g.Y : g.slabs + (size_t)zs * g.slab_stride);
    const __amdgpu_buffer_rsrc_t rsAdd = fd_make_rsrc(g.add ? g.add : g.Y);
    const bool has_add = final_pass && g.add, has_bias = final_pass && g.bias, relu = final_pass && g.act == 1;
    const unsigned cs4 = 4u * (unsigned)g.out_cs;
    const long p = p0 + wave * 32 + l31;
    unsigned pix = FD_OOB;
    if (p < Np) {
        const int n = (int)(p / plane);
        const int rem = (int)(p - (long)n * plane);
        const int y = rem / g.NX, x = rem - y * g.NX;
        pix = 4u * (unsigned)((long)n * g.out_ns + (long)(y * g.osy + g.ooy) * g.out_w + (x * g.osx + g.oox));
    }
#pragma unroll
    for (int i = 0; i < MB; ++i) {
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            unsigned off[4];
            float addv[4], bv[4];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int m = m0 + i * 32 + r4 + 8 * rq + 4 * half;
                off[r4] = m < g.M ? pix + (unsigned)m * cs4 : FD_OOB;
                bv[r4] = has_bias ? g.bias[m < g.M ? m : g.M - 1] : 0.f;
                addv[r4] = has_add ? fd_ldg32(rsAdd, off[r4]) : 0.f;
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                float v = acc[i][rq * 4 + r4] + bv[r4];
                v = relu ? (v > 0.f ? v : 0.f) : v;
                v += addv[r4];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, (int)off[r4], 0, 0);
            }
        }
    }
}

template <int MB, int D>
__global__ void __launch_bounds__(256, 2) k_conv_limb(FastGemmArgs g, int ntm, int ntn) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int slot = (int)blockIdx.x >> 3;
    const int tm = slot % ntm, tn = (slot / ntm) * 8 + ((int)blockIdx.x & 7);
    if (tn >= ntn) return;
    conv_limb_body<MB, D>(g, smem, tm, tn, (int)blockIdx.y, (int)gridDim.y);
}

// the output-parity classes of a stride-2 data gradient side by side (conv_fast.hip: k_conv_fast_grp); q.first_bx in units of this
// kernel's 1-D grid (8 * ceil(pixel tiles / 8) * ntm workgroups per class)
template <int MB, int D>
__global__ void __launch_bounds__(256, 2) k_conv_limb_grp(FastGemmArgs g, FastGemmGroup q, int ntm) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int j = 0;
    while (j + 1 < q.n && (int)blockIdx.x >= q.first_bx[j + 1]) ++j;
    g.A = q.A[j]; g.NY = q.NY[j]; g.NX = q.NX[j]; g.oy = q.oy[j]; g.ox = q.ox[j]; g.ooy = q.ooy[j]; g.oox = q.oox[j];
    g.T = q.T[j]; g.TB = q.TB[j]; g.K = q.K[j];
    if (q.own_out) { g.Y += q.y_off[j]; g.out_ns = q.out_ns[j]; g.out_cs = q.out_cs[j]; g.out_w = q.out_w[j]; }
    const int bx = (int)blockIdx.x - q.first_bx[j];
    const int ntn = (int)(((long)g.Nb * g.NY * g.NX + 127) / 128);
    const int slot = bx >> 3;
    const int tm = slot % ntm, tn = (slot / ntm) * 8 + (bx & 7);
    if (tn >= ntn) return;
    conv_limb_body<MB, D>(g, smem, tm, tn, 0, 1);
}

// A3 of the matrix A[m][kk = t * Cr + c] = W[co][ci][kh0 + dkh a][kw0 + dkw b], t = a * TB + b; mode 0: (m, c) = (co, ci) - forward;
// mode 1: (m, c) = (ci, co) - data gradient (the stand-alone form of re-layout modes 9 / 10)
__global__ void __launch_bounds__(256) k_limb_conv_weight_split(const float* __restrict__ W, uint4* __restrict__ A3, int Co, int Ci, int KH, int KW,
                                                                int TA, int TB, int kh0, int dkh, int kw0, int dkw, int mode) {
    const int Mr = mode ? Ci : Co, Cr = mode ? Co : Ci;
    const long K = (long)TA * TB * Cr;
    const long n = (long)Mr * (K >> 3);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int m = (int)(i % Mr);
        const long kk = (i / Mr) * 8;
        const int t = (int)(kk / Cr), c0 = (int)(kk - (long)t * Cr);
        const int a = t / TB, b = t - a * TB;
        const int kh = kh0 + dkh * a, kw = kw0 + dkw * b;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int co = mode ? c0 + e : m, ci = mode ? m : c0 + e;
            x[e] = W[(((long)co * Ci + ci) * KH + kh) * KW + kw];
        }
        uint4 h, md, l;
        split8(x, h, md, l);
        A3[a3_piece(kk, 0, m, Mr)] = h;
        A3[a3_piece(kk, 1, m, Mr)] = md;
        A3[a3_piece(kk, 2, m, Mr)] = l;
    }
}

// ------------------------------------------------------------------------------------------------ weight gradient
struct LimbWgradArgs {
    const float* dY;       // [Nb][M][HW]
    const float* X;        // [Nb][C][HW]
    float* slabs;          // [splits][M][C]
    int M, C, Nb, HW;
    int ntm, ntn;          // tiles over M (dY channels) and C (X channels)
    int chunks_per_split;  // K chunks of 32 pixels per split; the K axis is (image, pixel) flattened, HW % 8 == 0
};

// Workgroup = WAVES_M x WAVES_N waves, a wave owns 64 (dY channels) x 32 (X channels) outputs.  K (= pixels) runs in chunks of 32:
// a row's 32 pixels are one 128-byte line; a thread owns (row, octet) items - two 16-byte loads, one split, three 16-byte LDS
// stores.  Like the forward kernel this one is bound by memory latency, not by the matrix pipes (few K-chunks per workgroup, every
// byte read once): the loads of chunk c + D are issued behind the LDS stores of chunk c (register ring of D slots, statically
// indexed), so a chunk's data has D - 1 whole iterations to arrive.  LDS is single-buffered (two barriers per chunk; 2 - 4
// workgroups per CU overlap each other's store and MFMA phases).
template <int WAVES_M, int WAVES_N, int D>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_wgrad_limb(LimbWgradArgs g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = 64 * WAVES_M, BN = 32 * WAVES_N;
    constexpr int A_BYTES = 192 * BM;                                 // [2 k-steps][3 limbs][2 halves][rows][16 B]
    constexpr int NAI = BM * 4 / NT, NBI = BN * 4 / NT;               // (row, octet) items per thread
    static_assert((BM * 4) % NT == 0 && (BN * 4) % NT == 0, "loader mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [192 (BM + BN)]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int tm = (int)blockIdx.x % g.ntm, tn = (int)blockIdx.x / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const long Kall = (long)g.Nb * g.HW;
    const long k_lo = (long)blockIdx.y * g.chunks_per_split * 32;
    long k_hi = k_lo + (long)g.chunks_per_split * 32;
    k_hi = k_hi < Kall ? k_hi : Kall;

    // loader: a wave-instruction covers 16 consecutive rows x 4 octets (conflict-free 16-byte LDS stores: a store group of 8 lanes
    // = 8 consecutive rows of one octet); item i of a thread = (row (wave + NT/64 i) 16 + lane % 16, octet lane / 16)
    const int r16 = lane & 15, kq = lane >> 4;
    long kk = k_lo + 8 * kq;                                          // this thread's K index in the next chunk to load
    int n_img = (int)(kk / g.HW);
    int pix = (int)(kk - (long)n_img * g.HW);
    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.dY), rsB = fd_make_rsrc(g.X);
    unsigned a_row[NAI], b_row[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) { int m = m0 + (wave + (NT / 64) * i) * 16 + r16; m = m < g.M ? m : g.M - 1; a_row[i] = 4u * (unsigned)m * (unsigned)g.HW; }
#pragma unroll
    for (int i = 0; i < NBI; ++i) { int c = n0 + (wave + (NT / 64) * i) * 16 + r16; c = c < g.C ? c : g.C - 1; b_row[i] = 4u * (unsigned)c * (unsigned)g.HW; }
    float4 xa[D][NAI][2], xb[D][NBI][2];
    auto load_chunk = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const bool ok = kk < k_hi;
        const unsigned oa = ok ? 4u * ((unsigned)n_img * (unsigned)g.M * (unsigned)g.HW + (unsigned)pix) : FD_OOB;
        const unsigned ob = ok ? 4u * ((unsigned)n_img * (unsigned)g.C * (unsigned)g.HW + (unsigned)pix) : FD_OOB;
#pragma unroll
        for (int i = 0; i < NAI; ++i) { xa[S][i][0] = fd_ldg128(rsA, oa + a_row[i]); xa[S][i][1] = fd_ldg128(rsA, oa + a_row[i] + 16u); }
#pragma unroll
        for (int i = 0; i < NBI; ++i) { xb[S][i][0] = fd_ldg128(rsB, ob + b_row[i]); xb[S][i][1] = fd_ldg128(rsB, ob + b_row[i] + 16u); }
        kk += 32; pix += 32;
        while (pix >= g.HW) { pix -= g.HW; ++n_img; }
    };
    // piece of (k-step s = kq / 2, limb L, half kq % 2, row): ((s * 3 + L) * 2 + half) * ROWS + row
    auto split_store = [&](const float4& v0, const float4& v1, unsigned char* q, int limb_stride) __attribute__((always_inline)) {
        uint4 h, m, l;
        split2(v0.x, v0.y, h.x, m.x, l.x); split2(v0.z, v0.w, h.y, m.y, l.y);
        split2(v1.x, v1.y, h.z, m.z, l.z); split2(v1.z, v1.w, h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(q) = h;
        *reinterpret_cast<uint4*>(q + limb_stride) = m;
        *reinterpret_cast<uint4*>(q + 2 * limb_stride) = l;
    };
    auto store_chunk = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const int s6 = (kq >> 1) * 6 + (kq & 1);
#pragma unroll
        for (int i = 0; i < NAI; ++i)
            split_store(xa[S][i][0], xa[S][i][1], smem + 16 * (s6 * BM + (wave + (NT / 64) * i) * 16 + r16), 32 * BM);
#pragma unroll
        for (int i = 0; i < NBI; ++i)
            split_store(xb[S][i][0], xb[S][i][1], smem + A_BYTES + 16 * (s6 * BN + (wave + (NT / 64) * i) * 16 + r16), 32 * BN);
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    const int nchunk = (int)((k_hi - k_lo + 31) / 32);
    if (nchunk > 0) {
        static_for<D>([&](auto j) __attribute__((always_inline)) { load_chunk(j); });
        for (int c = 0; c < nchunk; c += D) {                  // branch-free body (see k_gemm_limb): chunks past the end are zeros
            static_for<D>([&](auto j) __attribute__((always_inline)) {
                store_chunk(j);
                __syncthreads();
                load_chunk(j);                                 // chunk c + J + D into the slot just emptied; in flight for D - 1 iterations
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned char* sa = smem + 16 * ((s * 6 + half) * BM + wave_m * 64 + l31);
                    const unsigned char* sb = smem + A_BYTES + 16 * ((s * 6 + half) * BN + wave_n * 32 + l31);
                    uint4 af[2][3], bf[3];
#pragma unroll
                    for (int L = 0; L < 3; ++L) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) af[i][L] = *reinterpret_cast<const uint4*>(sa + 32 * BM * L + 512 * i);
                        bf[L] = *reinterpret_cast<const uint4*>(sb + 32 * BN * L);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) FD_LIMB_MFMA6(acc[i], af[i], bf);
                }
                __syncthreads();
            });
        }
    }
    // ---- slab [z][m][c]: rows = dY channels, columns = X channels (contiguous)
    float* slab = g.slabs + (size_t)blockIdx.y * (size_t)g.M * g.C;
    const int c = n0 + wave_n * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < g.M && c < g.C) slab[(size_t)m * g.C + c] = acc[i][r];
        }
}

// ------------------------------------------------------------------------------------------------ weight gradient, 3x3 stride 2
// dW[m][c][ta][tb] = sum_{n, y, x} dY[n][m][y][x] X[n][c][2y + ta - 1][2x + tb - 1] (zero padding): k_wgrad_limb's arithmetic with one TAP per
// workgroup (grid x = tap x tile) and a loader built for the stride:
//   * a K-chunk = 32 consecutive output pixels = 8 groups of 4 (a group never leaves its output row: Wo % 4 == 0); the groups' (image, row,
//     column) are WAVE-UNIFORM and advance on the scalar unit - no per-lane index arithmetic at all;
//   * the X operand of a group = input pixels 2x .. 2x+7 of ONE input row (32 contiguous bytes); lanes are (row = lane / 4, kq = lane % 4),
//     load instruction j of an item reads 16-byte piece 4j + kq of the chunk's 256-byte window, so four adjacent lanes read 64 contiguous
//     bytes (the first version - a lane reads its own 8 pixels' 64 bytes, adjacent lanes = different channel rows - touched 6x the cache
//     lines per instruction and ran at 0.6x the f32 kernel: profiles/round6_limb_s2_ab.log).  A piece holds 2 output pixels' inputs: the
//     even ones for tb = 1, the odd ones for tb = 2, and for tb = 0 the odd ones of the window shifted by 8 bytes (the 16-byte buffer loads
//     need 4-byte alignment only), the pixel left of the image replaced by 0;
//   * so a lane's 8 K-values are the pixels {8j + 2kq, 8j + 2kq + 1 : j = 0..3} of the chunk - a permutation of the summation index, applied
//     to dY (four 8-byte loads per item) and X alike;
//   * LDS planes are padded by 32 bytes: the 16-byte stores of 8 adjacent lanes (2 rows x 4 kq -> planes 0, 1, 6, 7) fall into 8 distinct
//     16-byte slots of the 256-byte bank window.
// Needs W % 8 == 0 (Wo % 4 == 0, aligned pieces) and (Ho Wo) % 8 == 0; other shapes stay on k_wgrad_fast.
struct LimbWgradS2Args {
    const float* dY;       // [Nb][M][NY * NX]
    const float* X;        // [Nb][C][Hi][Wi]
    float* slabs;          // [splits][M][9][C]
    int M, C, Nb, Hi, Wi, NY, NX;
    int ntm, ntn;
    int chunks_per_split;
    int tap0, ntaps;       // 0, 9: the 3x3 kernel; 4, 1: a 1x1 stride-2 kernel without padding = the centre tap alone (slabs [splits][M][1][C])
};

template <int WAVES_M, int WAVES_N, int D>
__global__ void __launch_bounds__(64 * WAVES_M * WAVES_N) k_wgrad_limb_s2(LimbWgradS2Args g) {
    constexpr int NT = 64 * WAVES_M * WAVES_N;
    constexpr int BM = 64 * WAVES_M, BN = 32 * WAVES_N;
    constexpr int PA = 16 * BM + 32, PB = 16 * BN + 32;              // padded plane strides (bytes)
    constexpr int A_BYTES = 12 * PA;
    constexpr int NAI = BM * 4 / NT, NBI = BN * 4 / NT;
    static_assert((BM * 4) % NT == 0 && (BN * 4) % NT == 0, "loader mismatch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int tiles = g.ntm * g.ntn;
    const int ts = (int)blockIdx.x / tiles, tile = (int)blockIdx.x - ts * tiles;
    const int t = g.tap0 + ts;
    const int ta = t / 3, tb = t - 3 * ta;
    const int tm = tile % g.ntm, tn = tile / g.ntm;
    const int m0 = tm * BM, n0 = tn * BN;
    const int plane = g.NY * g.NX;
    const unsigned chw = (unsigned)(g.Hi * g.Wi);
    const long Kall = (long)g.Nb * plane;
    const long k_lo = (long)blockIdx.y * g.chunks_per_split * 32;
    long k_hi = k_lo + (long)g.chunks_per_split * 32;
    k_hi = k_hi < Kall ? k_hi : Kall;

    const int rl = lane >> 2, kq = lane & 3;
    // wave-uniform position of the next 4-pixel group to load: flat index kg, (image, output row, output column)
    long kg = k_lo;
    int gn = (int)(k_lo / plane);
    int gy, gx;
    { const int rem = (int)(k_lo - (long)gn * plane); gy = rem / g.NX; gx = rem - gy * g.NX; }
    const __amdgpu_buffer_rsrc_t rsA = fd_make_rsrc(g.dY), rsB = fd_make_rsrc(g.X);
    unsigned a_lane[NAI], b_lane[NBI];
#pragma unroll
    for (int i = 0; i < NAI; ++i) {
        int m = m0 + (wave + (NT / 64) * i) * 16 + rl; m = m < g.M ? m : g.M - 1;
        a_lane[i] = 4u * (unsigned)m * (unsigned)plane + 8u * (unsigned)kq;
    }
    const unsigned tb_shift = tb == 0 ? 8u : 0u;
#pragma unroll
    for (int i = 0; i < NBI; ++i) {
        int c = n0 + (wave + (NT / 64) * i) * 16 + rl; c = c < g.C ? c : g.C - 1;
        b_lane[i] = 4u * (unsigned)c * chw + 16u * (unsigned)(kq & 1);
    }
    f32x2_t xa[D][NAI][4];
    float4 xb[D][NBI][4];
    unsigned lz[D];                                                  // bit j: piece j of this lane starts at output column 0 (tb = 0: its first value is padding)
    auto load_chunk = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        unsigned zmask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // groups 2j and 2j + 1 (8 output pixels): dY offset of the pair, X offsets of each
            unsigned ob[2];
            unsigned oa = FD_OOB;
            bool x0[2], rok[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool ok = kg < k_hi;
                const int r = 2 * gy + ta - 1;
                const bool rowok = ok & ((unsigned)r < (unsigned)g.Hi);
                const unsigned base = 4u * ((unsigned)gn * (unsigned)g.C * chw + (unsigned)(r * g.Wi + 2 * gx));
                ob[q] = base; rok[q] = rowok;
                x0[q] = gx == 0;
                if (q == 0) oa = ok ? 4u * ((unsigned)gn * (unsigned)g.M * (unsigned)plane + (unsigned)(gy * g.NX + gx)) : FD_OOB;
                kg += 4; gx += 4;
                const bool w1 = gx >= g.NX;
                gx = w1 ? 0 : gx;
                gy += w1 ? 1 : 0;
                const bool w2 = gy >= g.NY;
                gy = w2 ? 0 : gy;
                gn += w2 ? 1 : 0;
            }
            // tb = 0: the window starts 8 bytes earlier (input columns 2x - 2 .. 2x + 1) - except where x = 0: that piece stays where it is
            // (no bytes in front of the tensor are touched) and its first value is the padding zero
            const bool zl = ((kq & 2) ? x0[1] : x0[0]) && (kq & 1) == 0;
            const unsigned obl = ((kq & 2) ? rok[1] : rok[0]) ? ((kq & 2) ? ob[1] : ob[0]) - (zl ? 0u : tb_shift) : FD_OOB;
            zmask |= zl ? (1u << j) : 0u;
#pragma unroll
            for (int i = 0; i < NAI; ++i)
                xa[S][i][j] = __builtin_bit_cast(f32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsA, (int)(oa + a_lane[i]), 0, 0));
#pragma unroll
            for (int i = 0; i < NBI; ++i) xb[S][i][j] = fd_ldg128(rsB, obl + b_lane[i]);
        }
        lz[S] = zmask;
    };
    auto split_store = [&](const float (&v)[8], unsigned char* q, int limb_stride) __attribute__((always_inline)) {
        uint4 h, m, l;
        split2(v[0], v[1], h.x, m.x, l.x); split2(v[2], v[3], h.y, m.y, l.y);
        split2(v[4], v[5], h.z, m.z, l.z); split2(v[6], v[7], h.w, m.w, l.w);
        *reinterpret_cast<uint4*>(q) = h;
        *reinterpret_cast<uint4*>(q + limb_stride) = m;
        *reinterpret_cast<uint4*>(q + 2 * limb_stride) = l;
    };
    auto store_chunk = [&](auto slot_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(slot_tag)::value;
        const int s6 = (kq >> 1) * 6 + (kq & 1);
#pragma unroll
        for (int i = 0; i < NAI; ++i) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = xa[S][i][j].x; v[2 * j + 1] = xa[S][i][j].y; }
            split_store(v, smem + s6 * PA + 16 * ((wave + (NT / 64) * i) * 16 + rl), 2 * PA);
        }
#pragma unroll
        for (int i = 0; i < NBI; ++i) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 f = xb[S][i][j];
                const bool zl = tb == 0 && ((lz[S] >> j) & 1u);
                v[2 * j] = zl ? 0.f : (tb == 1 ? f.x : f.y);
                v[2 * j + 1] = tb == 1 ? f.z : (zl ? f.y : f.w);
            }
            split_store(v, smem + A_BYTES + s6 * PB + 16 * ((wave + (NT / 64) * i) * 16 + rl), 2 * PB);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int half = lane >> 5, l31 = lane & 31;
    const int nchunk = (int)((k_hi - k_lo + 31) / 32);
    if (nchunk > 0) {
        static_for<D>([&](auto j) __attribute__((always_inline)) { load_chunk(j); });
        for (int c = 0; c < nchunk; c += D) {
            static_for<D>([&](auto j) __attribute__((always_inline)) {
                store_chunk(j);
                __syncthreads();
                load_chunk(j);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const unsigned char* sa = smem + (s * 6 + half) * PA + 16 * (wave_m * 64 + l31);
                    const unsigned char* sb = smem + A_BYTES + (s * 6 + half) * PB + 16 * (wave_n * 32 + l31);
                    uint4 af[2][3], bf[3];
#pragma unroll
                    for (int L = 0; L < 3; ++L) {
#pragma unroll
                        for (int i = 0; i < 2; ++i) af[i][L] = *reinterpret_cast<const uint4*>(sa + 2 * PA * L + 512 * i);
                        bf[L] = *reinterpret_cast<const uint4*>(sb + 2 * PB * L);
                    }
#pragma unroll
                    for (int i = 0; i < 2; ++i) FD_LIMB_MFMA6(acc[i], af[i], bf);
                }
                __syncthreads();
            });
        }
    }
    // ---- slab [z][m][t][c]
    float* slab = g.slabs + (size_t)blockIdx.y * (size_t)g.M * g.ntaps * g.C;
    const int c = n0 + wave_n * 32 + l31;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave_m * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (m < g.M && c < g.C) slab[((size_t)m * g.ntaps + ts) * g.C + c] = acc[i][r];
        }
}

// ------------------------------------------------------------------------------------------------ weight pre-split (stand-alone)
// A3 piece (m, kk .. kk+7) from W: transposed == 0: A[m][k] = W[m * K + k] (forward: m = Cout, k = Cin);
// transposed != 0: A[m][k] = W[k * M + m] (data gradient: m = Cin, k = Cout)
__global__ void __launch_bounds__(256) k_limb_weight_split(const float* __restrict__ W, uint4* __restrict__ A3, int M, int K, int transposed) {
    const long n = (long)M * (K >> 3);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int m = (int)(i % M);
        const long kk = (i / M) * 8;
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = transposed ? W[(kk + e) * M + m] : W[(long)m * K + kk + e];
        uint4 h, md, l;
        split8(x, h, md, l);
        A3[a3_piece(kk, 0, m, M)] = h;
        A3[a3_piece(kk, 1, m, M)] = md;
        A3[a3_piece(kk, 2, m, M)] = l;
    }
}

inline int limb_splits(long tiles, int nch, int target) {
    if (tiles >= target || nch < 8) return 1;
    long want = (target + tiles - 1) / tiles;
    const long maxs = nch / 4;                        // at least 4 K-chunks (64 channels) per split
    if (want > maxs) want = maxs;
    if (want < 1) want = 1;
    if (want > 16) want = 16;
    const int per = (int)((nch + want - 1) / want);
    return (nch + per - 1) / per;                     // no empty split
}
}  // namespace

// ---- interface to conv.hip -------------------------------------------------------------------------------------------
bool limb_1x1_shape(const fd_conv_desc* d) {
    return d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 && !d->in_norm;
}
bool limb_fwd_ok(const fd_conv_desc* d) {
    return fd_tun().limb_1x1 != 0 && limb_1x1_shape(d) && d->Cin % 32 == 0 && d->Cin >= 64 && d->Cout >= 64 && (d->act == 0 || d->act == 1);
}
bool limb_dgrad_ok(const fd_conv_desc* d) {
    return fd_tun().limb_1x1 != 0 && limb_1x1_shape(d) && d->Cout % 32 == 0 && d->Cout >= 64 && d->Cin >= 64;
}
bool limb_wgrad_ok(const fd_conv_desc* d) {
    // (64 -> 64 stays on the f32 kernel: one 128 x 64 tile per slice, 38 vs 47 us at 48x160 - profiles/round6_limb_ab_v5_sweep.log)
    return fd_tun().limb_1x1 != 0 && limb_1x1_shape(d) && ((long)d->H * d->W) % 8 == 0 && d->Cin >= 64 && d->Cout >= 64 && d->Cin + d->Cout > 128;
}
long limb_wt_floats(long M, long K) { return a3_floats(M, K); }

namespace {
struct LimbCfg { int wm, wn; };
inline int limb_gemm_mb(int M) { return M <= 64 ? 2 : 4; }          // 32-row blocks per workgroup tile: 64 or 128 channels
inline int limb_gemm_depth(int K) {                                   // the unrolled loop needs K % (16 D) == 0
    const int want = fd_tun().limb_depth >= 4 ? 4 : 2;
    return (K % (16 * want) == 0) ? want : 2;
}
}  // namespace

int limb_gemm_splits(int M, int K, int Nb, int HW) {
    const long tiles = (long)fd_cdiv(M, 32 * limb_gemm_mb(M)) * fd_cdiv((long)Nb * HW, 128);
    // a split costs a slab write + read of the whole output: only for the deep layers (few tiles, small planes, long K)
    if ((long)Nb * M * HW > (long)fd_tun().limb_split_max_out) return 1;
    const int nch = K >> 4, D = limb_gemm_depth(K);
    int sp = limb_splits(tiles, nch, fd_tun().limb_target);
    while (sp > 1 && (nch % sp != 0 || (nch / sp) % D != 0)) --sp;   // equal shares, each a multiple of the unroll depth
    return sp;
}
long limb_gemm_ws_floats(int M, int K, int Nb, int HW) {
    const int sp = limb_gemm_splits(M, K, Nb, HW);
    return sp > 1 ? (long)sp * Nb * M * HW : 0;
}

int limb_weight_split_launch(const float* w, float* wt, int M, int K, int transposed, hipStream_t st) {
    const long n = (long)M * (K >> 3);
    long b = (n + 255) / 256;
    b = b > 2048 ? 2048 : b;
    hipLaunchKernelGGL(k_limb_weight_split, dim3((unsigned)b), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wt), M, K, transposed);
    FD_LAUNCH_CHECK("limb weight split");
    return 0;
}

// Y[n][m][p] = act(sum_k A[m][k] X[n][k][p] + bias[m]) + add[n][m][p]
int limb_gemm_launch(const float* wt, const float* x, float* y, const float* bias, const float* add, float* ws, int M, int K, int Nb, int HW,
                     int act, hipStream_t st) {
    const int mb = limb_gemm_mb(M);
    LimbGemmArgs g = {};
    g.A3 = wt; g.X = x; g.Y = y; g.bias = bias; g.add = add;
    g.M = M; g.K = K; g.Nb = Nb; g.HW = HW; g.act = act;
    g.ntm = fd_cdiv(M, 32 * mb); g.ntn = fd_cdiv((long)Nb * HW, 128);
    const int splits = limb_gemm_splits(M, K, Nb, HW);
    g.slab_stride = (long)Nb * M * HW;
    g.slabs = ws;
    FD_REQUIRE(K % 32 == 0, "limb gemm: K must be a multiple of 32");
    FD_REQUIRE(splits == 1 || ws, "limb gemm: split-K workspace missing");
    const dim3 grid(8u * (unsigned)fd_cdiv(g.ntn, 8) * (unsigned)g.ntm, (unsigned)splits);
    const size_t lds = 2 * 96 * (size_t)(32 * mb);
    const int depth = limb_gemm_depth(K);
    if (mb == 2) {
        if (depth == 2) hipLaunchKernelGGL((k_gemm_limb<2, 2>), grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL((k_gemm_limb<2, 4>), grid, dim3(256), lds, st, g);
    } else {
        if (depth == 2) hipLaunchKernelGGL((k_gemm_limb<4, 2>), grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL((k_gemm_limb<4, 4>), grid, dim3(256), lds, st, g);
    }
    FD_LAUNCH_CHECK("limb gemm");
    if (splits > 1)
        return fast_splitk_finish_launch(ws, y, bias, g.slab_stride, g.slab_stride, splits, HW, M, act, st, add);
    return 0;
}

// ---- implicit-GEMM convolutions (k_conv_limb): a problem in conv_fast.h's terms, a.A = the A3 image of [M][(tap, channel)]
bool limb_conv_problem_ok(int M, int C, int pad_mode, int act) {
    return fd_tun().limb_conv != 0 && C % 32 == 0 && C >= 64 && M >= 64 && pad_mode == 0 && (act == 0 || act == 1);
}
int limb_conv_weight_split_launch(const float* w, float* wt, int Co, int Ci, int KH, int KW, int TA, int TB, int kh0, int dkh, int kw0, int dkw,
                                  int mode, hipStream_t st) {
    const long n = (long)(mode ? Ci : Co) * (((long)TA * TB * (mode ? Co : Ci)) >> 3);
    long b = (n + 255) / 256;
    b = b > 2048 ? 2048 : b;
    hipLaunchKernelGGL(k_limb_conv_weight_split, dim3((unsigned)b), dim3(256), 0, st, w, reinterpret_cast<uint4*>(wt), Co, Ci, KH, KW, TA, TB,
                       kh0, dkh, kw0, dkw, mode);
    FD_LAUNCH_CHECK("limb conv weight split");
    return 0;
}
int limb_conv_splits(const FastGemmArgs& a) {
    if (!(a.osy == 1 && a.osx == 1) || a.out_total > (long)fd_tun().limb_split_max_out) return 1;
    const long tiles = (long)fd_cdiv(a.M, 32 * limb_gemm_mb(a.M)) * fd_cdiv((long)a.Nb * a.NY * a.NX, 128);
    const int nch = a.T * (a.C >> 4);
    int sp = limb_splits(tiles, nch, fd_tun().limb_target);
    while (sp > 1 && (nch % sp != 0 || (nch / sp) % 2 != 0)) --sp;
    return sp;
}
long limb_conv_ws_floats(const FastGemmArgs& a) {
    const int sp = limb_conv_splits(a);
    return sp > 1 ? (long)sp * a.out_total : 0;
}
int limb_conv_launch(const FastGemmArgs& a, hipStream_t st) {
    if ((double)a.Nb * a.C * a.Hi * a.Wi * 4.0 >= 2147483648.0 || (double)a.M * a.K * 6.0 >= 2147483648.0 ||
        (double)a.out_total * 4.0 >= 2147483648.0) {
        fd_set_error("limb conv: tensor exceeds the 2 GiB addressing range"); return -1;
    }
    FD_REQUIRE(a.C % 32 == 0 && a.K == a.T * a.C, "limb conv: channels must be a multiple of 32");
    const int mb = limb_gemm_mb(a.M);
    const int ntm = fd_cdiv(a.M, 32 * mb), ntn = fd_cdiv((long)a.Nb * a.NY * a.NX, 128);
    const int splits = limb_conv_splits(a);
    FD_REQUIRE(splits == 1 || a.slabs, "limb conv: split-K workspace missing");
    const dim3 grid(8u * (unsigned)fd_cdiv(ntn, 8) * (unsigned)ntm, (unsigned)splits);
    const size_t lds = 2 * 96 * (size_t)(32 * mb);
    if (mb == 2) hipLaunchKernelGGL((k_conv_limb<2, 2>), grid, dim3(256), lds, st, a, ntm, ntn);
    else hipLaunchKernelGGL((k_conv_limb<4, 2>), grid, dim3(256), lds, st, a, ntm, ntn);
    FD_LAUNCH_CHECK("limb conv");
    if (splits > 1)
        return fast_splitk_finish_launch(a.slabs, a.Y, a.bias, a.out_total, a.slab_stride, splits, a.out_cs, a.M, a.act, st, a.add);
    return 0;
}
int limb_conv_group_launch(const FastGemmArgs& a, const FastGemmGroup& q, hipStream_t st) {
    if ((double)a.Nb * a.C * a.Hi * a.Wi * 4.0 >= 2147483648.0 || (double)a.out_total * 4.0 >= 2147483648.0) {
        fd_set_error("limb conv: tensor exceeds the 2 GiB addressing range"); return -1;
    }
    FD_REQUIRE(a.C % 32 == 0, "limb conv: channels must be a multiple of 32");
    const int mb = limb_gemm_mb(a.M);
    const int ntm = fd_cdiv(a.M, 32 * mb);
    FastGemmGroup grp = q;
    int gx = 0;
    for (int j = 0; j < q.n; ++j) {
        if ((double)a.M * q.K[j] * 6.0 >= 2147483648.0) { fd_set_error("limb conv: weights exceed the 2 GiB addressing range"); return -1; }
        grp.first_bx[j] = gx;
        gx += 8 * fd_cdiv(fd_cdiv((long)a.Nb * q.NY[j] * q.NX[j], 128), 8) * ntm;
    }
    grp.first_bx[q.n] = gx;
    const size_t lds = 2 * 96 * (size_t)(32 * mb);
    if (mb == 2) hipLaunchKernelGGL((k_conv_limb_grp<2, 2>), dim3((unsigned)gx), dim3(256), lds, st, a, grp, ntm);
    else hipLaunchKernelGGL((k_conv_limb_grp<4, 2>), dim3((unsigned)gx), dim3(256), lds, st, a, grp, ntm);
    FD_LAUNCH_CHECK("limb conv (grouped)");
    return 0;
}

namespace {
inline LimbCfg limb_wgrad_cfg(int M, int C) {       // waves along M (64 dY channels each) x waves along C (32 X channels each)
    if (C <= 64) return LimbCfg{2, 2};
    if (M <= 64) return LimbCfg{1, 4};
    return LimbCfg{2, 4};
}
}  // namespace
int limb_wgrad_splits(int M, int C, int Nb, int HW, int* chunks_per_split) {
    const LimbCfg c = limb_wgrad_cfg(M, C);
    const long tiles = (long)fd_cdiv(M, 64 * c.wm) * fd_cdiv(C, 32 * c.wn);
    const long nch = ((long)Nb * HW + 31) / 32;
    const long target = (long)fd_tun().limb_wgrad_target * (c.wm * c.wn == 8 ? 1 : 2);      // 4-wave workgroups: twice as many fit a CU
    long want = (target + tiles - 1) / tiles;
    long maxs = nch / 4;                                  // at least 4 chunks (128 pixels) per split
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    const long per = (nch + want - 1) / want;
    if (chunks_per_split) *chunks_per_split = (int)per;
    return (int)((nch + per - 1) / per);
}
long limb_wgrad_ws_floats(int M, int C, int Nb, int HW) { return (long)limb_wgrad_splits(M, C, Nb, HW, nullptr) * M * C; }

int limb_wgrad_launch(const float* x, const float* gy, float* gw, float* ws, int M, int C, int Nb, int HW, int accumulate, hipStream_t st) {
    const LimbCfg c = limb_wgrad_cfg(M, C);
    LimbWgradArgs g = {};
    g.dY = gy; g.X = x; g.slabs = ws; g.M = M; g.C = C; g.Nb = Nb; g.HW = HW;
    g.ntm = fd_cdiv(M, 64 * c.wm); g.ntn = fd_cdiv(C, 32 * c.wn);
    const int splits = limb_wgrad_splits(M, C, Nb, HW, &g.chunks_per_split);
    const dim3 grid((unsigned)(g.ntm * g.ntn), (unsigned)splits);
    const size_t lds = 192 * (size_t)(64 * c.wm + 32 * c.wn);
    const bool deep = fd_tun().limb_depth >= 3;
    if (c.wm == 2 && c.wn == 2) {
        if (deep) hipLaunchKernelGGL((k_wgrad_limb<2, 2, 3>), grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL((k_wgrad_limb<2, 2, 2>), grid, dim3(256), lds, st, g);
    } else if (c.wm == 1) {
        if (deep) hipLaunchKernelGGL((k_wgrad_limb<1, 4, 3>), grid, dim3(256), lds, st, g);
        else hipLaunchKernelGGL((k_wgrad_limb<1, 4, 2>), grid, dim3(256), lds, st, g);
    } else {
        if (fd_tun().limb_depth >= 4) hipLaunchKernelGGL((k_wgrad_limb<2, 4, 3>), grid, dim3(512), lds, st, g);    // 134 registers: one workgroup per CU
        else hipLaunchKernelGGL((k_wgrad_limb<2, 4, 2>), grid, dim3(512), lds, st, g);
    }
    FD_LAUNCH_CHECK("limb wgrad");
    return fast_wgrad_finish_launch(ws, gw, M, C, 1, splits, accumulate, st);
}

// ---- 3x3 stride-2 weight gradient (k_wgrad_limb_s2)
bool limb_wgrad_s2_shape_ok(int M, int C, int Hi, int Wi, int NY, int NX) {
    return fd_tun().limb_conv != 0 && M >= 64 && C >= 64 && C % 32 == 0 && Wi % 8 == 0 && NX * 2 == Wi && NY * 2 >= Hi && NY * 2 <= Hi + 1 &&
           ((long)NY * NX) % 8 == 0;
}
int limb_wgrad_s2_splits(int M, int C, int Nb, int plane, int ntaps, int* chunks_per_split) {
    const LimbCfg c = limb_wgrad_cfg(M, C);
    const long tiles = (long)ntaps * fd_cdiv(M, 64 * c.wm) * fd_cdiv(C, 32 * c.wn);
    const long nch = ((long)Nb * plane + 31) / 32;
    const long target = (long)fd_tun().limb_wgrad_target * (c.wm * c.wn == 8 ? 1 : 2);
    long want = (target + tiles - 1) / tiles;
    long maxs = nch / 4;
    if (maxs < 1) maxs = 1;
    if (want > maxs) want = maxs;
    if (want > 512) want = 512;
    if (want < 1) want = 1;
    const long per = (nch + want - 1) / want;
    if (chunks_per_split) *chunks_per_split = (int)per;
    return (int)((nch + per - 1) / per);
}
long limb_wgrad_s2_ws_floats(int M, int C, int Nb, int plane, int ntaps) {
    return (long)limb_wgrad_s2_splits(M, C, Nb, plane, ntaps, nullptr) * M * ntaps * C;
}

// ntaps = 9: 3x3 kernel with padding 1; ntaps = 1: 1x1 kernel without padding (the centre tap's access pattern)
int limb_wgrad_s2_launch(const float* x, const float* gy, float* gw, float* ws, int M, int C, int Nb, int Hi, int Wi, int NY, int NX, int ntaps,
                         int accumulate, hipStream_t st) {
    const LimbCfg c = limb_wgrad_cfg(M, C);
    LimbWgradS2Args g = {};
    g.dY = gy; g.X = x; g.slabs = ws; g.M = M; g.C = C; g.Nb = Nb; g.Hi = Hi; g.Wi = Wi; g.NY = NY; g.NX = NX;
    g.ntm = fd_cdiv(M, 64 * c.wm); g.ntn = fd_cdiv(C, 32 * c.wn);
    g.ntaps = ntaps; g.tap0 = ntaps == 9 ? 0 : 4;
    const int splits = limb_wgrad_s2_splits(M, C, Nb, NY * NX, ntaps, &g.chunks_per_split);
    const dim3 grid((unsigned)(ntaps * g.ntm * g.ntn), (unsigned)splits);
    const size_t lds = 12 * (size_t)(16 * (64 * c.wm + 32 * c.wn) + 64);
    if (c.wm == 2 && c.wn == 2) hipLaunchKernelGGL((k_wgrad_limb_s2<2, 2, 2>), grid, dim3(256), lds, st, g);
    else if (c.wm == 1) hipLaunchKernelGGL((k_wgrad_limb_s2<1, 4, 2>), grid, dim3(256), lds, st, g);
    else hipLaunchKernelGGL((k_wgrad_limb_s2<2, 4, 2>), grid, dim3(512), lds, st, g);
    FD_LAUNCH_CHECK("limb wgrad (3x3 stride 2)");
    return fast_wgrad_finish_launch(ws, gw, M, C, ntaps, splits, accumulate, st);
}

// 1x1 stride-2 layers (the downsample branch of layerN.0): launch-bound GEMMs when the plane is small and K = Cin short - the ResNet-18
// ones at 640x192 are 11 - 26 us on either arithmetic and the limb kernel's split-K finish makes them slower; taken only when the launch
// fills the chip without a split (ResNet-50's: K = 256 .. 1024 at 2 - 4x the pixels)
bool limb_conv_1x1_worth(int M, long Np) { return (long)fd_cdiv(M, 32 * limb_gemm_mb(M)) * fd_cdiv(Np, 128) >= 200; }
