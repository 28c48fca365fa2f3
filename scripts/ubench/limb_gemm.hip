// Exploration for a split-precision convolution: what does an fp32-accurate GEMM cost on gfx950 when every
// fp32 operand is split into three bf16 limbs (x = h + m + l, 8 + 8 + 8 mantissa bits) inside the loader and the product is formed
// from the six limb products of weight >= 2^-16 (hh, hm, mh, mm, hl, lh) on the bf16 MFMA (v_mfma_f32_32x32x16_bf16, 16x the
// f32 MFMA rate, fp32 accumulation) - against the same tiling on the f32 MFMA (v_mfma_f32_32x32x2_f32)?
//
//   C[M][N] = A[M][K] * Bt[N][K]^T     (both operands K-contiguous, like the (tap, channel)-major operands of the conv kernels)
//
// Workgroup = 256 threads = 2 x 2 waves, 128 x 128 tile, 64 x 64 per wave, K chunks of 32, LDS single-buffered with the next chunk's
// global loads in flight during the MFMA phase.  Reports TFLOP/s of ALGORITHMIC flops (2 M N K) and the error against float64.
//
//   hipcc --offload-arch=gfx950 -O3 -o limb_gemm limb_gemm.hip && ./limb_gemm
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32, NT = 256;
constexpr int LDK = BK + 8;            // bf16 per LDS row: 80 bytes, keeps every 8-element fragment 16-byte aligned

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)x;
    const float r = x - (float)h;      // exact: h holds the leading 8 bits of x
    m = (__bf16)r;
    const float r2 = r - (float)m;     // exact
    l = (__bf16)r2;
}

__global__ void k_split(const float* __restrict__ x, __bf16* __restrict__ h, __bf16* __restrict__ m, __bf16* __restrict__ l, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) split3(x[i], h[i], m[i], l[i]);
}

// PRE_A: the A operand (the weights of a convolution: re-laid-out once per optimiser step anyway) arrives already split, as three
// bf16 matrices; only the activation operand is split in the loader.
template <int NPROD, bool PRE_A = false, bool PRE_B = false>   // 6: fp32-accurate; 3: hh + hm + mh (~2^-16); 1: plain bf16
__global__ void __launch_bounds__(NT) k_gemm_limb(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                  int M, int N, int K, const __bf16* __restrict__ A3 = nullptr,
                                                  const __bf16* __restrict__ B3 = nullptr) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3][BM][LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3][BN][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 31, kh = lane >> 5;

    constexpr int NLA = NPROD > 3 ? 3 : (NPROD > 1 ? 2 : 1);
    float4 ra[4], rb[4];
    bf16x8 pa[NLA][2];                 // PRE_A: 128 rows x 32 k of bf16 per limb = 512 fragments of 8 = 2 per thread
    bf16x8 pb[NLA][2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * NT, row = idx >> 3, kq = (idx & 7) * 4;
            if (!PRE_A) ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + k0 + kq);
            if (!PRE_B) rb[p] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * K + k0 + kq);
        }
        if (PRE_B) {
#pragma unroll
            for (int t = 0; t < NLA; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 8;
                    pb[t][p] = *reinterpret_cast<const bf16x8*>(B3 + (size_t)t * N * K + (size_t)(n0 + row) * K + k0 + kq);
                }
        }
        if (PRE_A) {
#pragma unroll
            for (int t = 0; t < NLA; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 8;
                    pa[t][p] = *reinterpret_cast<const bf16x8*>(A3 + (size_t)t * M * K + (size_t)(m0 + row) * K + k0 + kq);
                }
        }
    };
    auto lstore = [&]() {
        if (PRE_B) {
#pragma unroll
            for (int t = 0; t < NLA; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 8;
                    *reinterpret_cast<bf16x8*>(&sB[t][row][kq]) = pb[t][p];
                }
        }
        if (PRE_A) {
#pragma unroll
            for (int t = 0; t < NLA; ++t)
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 8;
                    *reinterpret_cast<bf16x8*>(&sA[t][row][kq]) = pa[t][p];
                }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (PRE_A && PRE_B) break;
            const int idx = tid + p * NT, row = idx >> 3, kq = (idx & 7) * 4;
            const float va[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w}, vb[4] = {rb[p].x, rb[p].y, rb[p].z, rb[p].w};
            bf16x4 ah, am, al, bh, bm, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __bf16 h, m, l;
                if (!PRE_A) { split3(va[e], h, m, l); ah[e] = h; am[e] = m; al[e] = l; }
                if (!PRE_B) { split3(vb[e], h, m, l); bh[e] = h; bm[e] = m; bl[e] = l; }
            }
            if (!PRE_A) *reinterpret_cast<bf16x4*>(&sA[0][row][kq]) = ah;
            if (!PRE_B) *reinterpret_cast<bf16x4*>(&sB[0][row][kq]) = bh;
            if (NPROD > 1) {
                if (!PRE_A) *reinterpret_cast<bf16x4*>(&sA[1][row][kq]) = am;
                if (!PRE_B) *reinterpret_cast<bf16x4*>(&sB[1][row][kq]) = bm;
            }
            if (NPROD > 3) {
                if (!PRE_A) *reinterpret_cast<bf16x4*>(&sA[2][row][kq]) = al;
                if (!PRE_B) *reinterpret_cast<bf16x4*>(&sB[2][row][kq]) = bl;
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            constexpr int NL = NPROD > 3 ? 3 : (NPROD > 1 ? 2 : 1);
            bf16x8 a[NL][2], b[NL][2];
#pragma unroll
            for (int t = 0; t < NL; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[t][i] = *reinterpret_cast<const bf16x8*>(&sA[t][wm * 64 + i * 32 + r][kb * 16 + kh * 8]);
                    b[t][i] = *reinterpret_cast<const bf16x8*>(&sB[t][wn * 64 + i * 32 + r][kb * 16 + kh * 8]);
                }
            // product-major order: consecutive MFMAs go to different accumulators (a chain of six on one accumulator would wait
            // for each result), smallest terms first
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int q = (NPROD > 3 ? 0 : (NPROD > 1 ? 3 : 5)); q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q] < NL ? PA[q] : 0][i], b[PB[q] < NL ? PB[q] : 0][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + r;
                C[(size_t)m * N + n] = acc[i][j][q];
            }
}

// Software-pipelined variant of the six-product kernel: K chunks of 16 (one MFMA K), LDS double-buffered (one barrier per chunk),
// and the loader woven between the MFMA groups of the wave itself: while the 24 MFMAs of chunk c run from buffer c % 2, the
// registers holding chunk c+1 are split and stored into the other buffer (after the first two product groups) and the global loads
// of chunk c+2 are issued (after the fourth).  Products hh first, so the first MFMAs need only two of the six fragment pairs.
constexpr int BK2 = 16, LDK2 = BK2 + 8;            // 48-byte rows: 16 lanes x 16-byte fragments cover the 64 banks exactly once
__global__ void __launch_bounds__(NT) k_gemm_limb_pipe(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                       int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][BM][LDK2];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][BN][LDK2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 31, kh = lane >> 5;
    // 128 rows x 16 k = 512 float4 per operand: two per thread
    float4 ra[2], rb[2];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
            ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + k0 + kq);
            rb[p] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * K + k0 + kq);
        }
    };
    auto lstore = [&](int buf, int p) __attribute__((always_inline)) {
        const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
        const float va[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w}, vb[4] = {rb[p].x, rb[p].y, rb[p].z, rb[p].w};
        bf16x4 qa[3], qb[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, m, l;
            split3(va[e], h, m, l); qa[0][e] = h; qa[1][e] = m; qa[2][e] = l;
            split3(vb[e], h, m, l); qb[0][e] = h; qb[1][e] = m; qb[2][e] = l;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *reinterpret_cast<bf16x4*>(&sA[buf][t][row][kq]) = qa[t];
            *reinterpret_cast<bf16x4*>(&sB[buf][t][row][kq]) = qb[t];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int nchunk = K / BK2;
    gload(0);
    lstore(0, 0); lstore(0, 1);
    if (nchunk > 1) gload(BK2);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        bf16x8 a[3][2], b[3][2];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[t][i] = *reinterpret_cast<const bf16x8*>(&sA[cur][t][wm * 64 + i * 32 + r][kh * 8]);
                b[t][i] = *reinterpret_cast<const bf16x8*>(&sB[cur][t][wn * 64 + i * 32 + r][kh * 8]);
            }
        constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};       // hh hm mh mm hl lh
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < nchunk) {
                if (q == 1) lstore(cur ^ 1, 0);
                if (q == 2) lstore(cur ^ 1, 1);
                if (q == 4 && c + 2 < nchunk) gload((c + 2) * BK2);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + r;
                C[(size_t)m * N + n] = acc[i][j][q];
            }
}

// Second pipelined variant: the fragments of chunk c+1 are read from LDS into a second register set DURING the last three product
// groups of chunk c (their latency hides behind 12 MFMAs), so the only exposed step per chunk is one barrier in its middle:
//   first half : 3 product groups on F(c)  +  split / store of chunk c+1 into the other buffer
//   barrier
//   second half: 3 product groups on F(c)  +  fragment reads F(c+1)  +  global loads of chunk c+2
__global__ void __launch_bounds__(NT) k_gemm_limb_pipe2(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                        int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][BM][LDK2];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][BN][LDK2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 31, kh = lane >> 5;
    float4 ra[2], rb[2];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
            ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + k0 + kq);
            rb[p] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * K + k0 + kq);
        }
    };
    auto lstore = [&](int buf, int p) __attribute__((always_inline)) {
        const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
        const float va[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w}, vb[4] = {rb[p].x, rb[p].y, rb[p].z, rb[p].w};
        bf16x4 qa[3], qb[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, m, l;
            split3(va[e], h, m, l); qa[0][e] = h; qa[1][e] = m; qa[2][e] = l;
            split3(vb[e], h, m, l); qb[0][e] = h; qb[1][e] = m; qb[2][e] = l;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *reinterpret_cast<bf16x4*>(&sA[buf][t][row][kq]) = qa[t];
            *reinterpret_cast<bf16x4*>(&sB[buf][t][row][kq]) = qb[t];
        }
    };
    bf16x8 fa[2][3][2], fb[2][3][2];
    auto fread = [&](int set, int buf, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            fa[set][t][i] = *reinterpret_cast<const bf16x8*>(&sA[buf][t][wm * 64 + i * 32 + r][kh * 8]);
            fb[set][t][i] = *reinterpret_cast<const bf16x8*>(&sB[buf][t][wn * 64 + i * 32 + r][kh * 8]);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};       // hh hm mh mm hl lh
    auto group = [&](int set, int q) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][PA[q]][i], fb[set][PB[q]][j], acc[i][j], 0, 0, 0);
    };
    const int nchunk = K / BK2;
    gload(0);
    lstore(0, 0); lstore(0, 1);
    if (nchunk > 1) gload(BK2);
    __syncthreads();
    fread(0, 0, 0); fread(0, 0, 1); fread(0, 0, 2);
    auto chunk = [&](int c, const int set) __attribute__((always_inline)) {      // `set` = c & 1, compile-time after unrolling by two
        const bool more = c + 1 < nchunk;
        group(set, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) lstore(set ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(set, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) lstore(set ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        group(set, 2);
        __syncthreads();
        if (more) fread(set ^ 1, set ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        group(set, 3);
        __builtin_amdgcn_sched_barrier(0);
        if (more) fread(set ^ 1, set ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        group(set, 4);
        __builtin_amdgcn_sched_barrier(0);
        if (more) fread(set ^ 1, set ^ 1, 2);
        if (c + 2 < nchunk) gload((c + 2) * BK2);
        __builtin_amdgcn_sched_barrier(0);
        group(set, 5);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int c = 0; c < nchunk; c += 2) {
        chunk(c, 0);
        if (c + 1 < nchunk) chunk(c + 1, 1);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + r;
                C[(size_t)m * N + n] = acc[i][j][q];
            }
}

// Large-tile variant, shaped by the LDS wall measured in mfma_bf16_lds.hip: 128 x 128 per WAVE (4 x 4 blocks, 256 accumulator
// registers, one wave per SIMD), workgroup = 2 x 2 waves = 256 x 256, K chunks of 16, LDS double-buffered (147 KB).  24 fragment reads
// feed 96 MFMAs (half the LDS bytes per MFMA of the 64 x 64 tile); fragments are requested in the order the product groups need them
// (hh -> a0 b0, hm -> b1, mh -> a1, mm, hl -> b2, lh -> a2) and the next chunk's split / store / global loads ride between the groups.
constexpr int BM3 = 256, BN3 = 256;
__global__ void __launch_bounds__(NT, 1) k_gemm_limb_big(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                         int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) __bf16 smem3[];
    // [buf][operand][limb][256 rows][LDK2]
    auto S = [&](int buf, int op, int t, int row, int k) -> __bf16* { return smem3 + ((((size_t)buf * 2 + op) * 3 + t) * 256 + row) * LDK2 + k; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM3, n0 = blockIdx.x * BN3;
    const int r = lane & 31, kh = lane >> 5;
    // 256 rows x 16 k = 1024 float4 per operand: four per thread
    float4 ra[4], rb[4];
    auto gload = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
            ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + k0 + kq);
            rb[p] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * K + k0 + kq);
        }
    };
    auto lstore = [&](int buf, int p) __attribute__((always_inline)) {
        const int idx = tid + p * NT, row = idx >> 2, kq = (idx & 3) * 4;
        const float va[4] = {ra[p].x, ra[p].y, ra[p].z, ra[p].w}, vb[4] = {rb[p].x, rb[p].y, rb[p].z, rb[p].w};
        bf16x4 qa[3], qb[3];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 h, m, l;
            split3(va[e], h, m, l); qa[0][e] = h; qa[1][e] = m; qa[2][e] = l;
            split3(vb[e], h, m, l); qb[0][e] = h; qb[1][e] = m; qb[2][e] = l;
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            *reinterpret_cast<bf16x4*>(S(buf, 0, t, row, kq)) = qa[t];
            *reinterpret_cast<bf16x4*>(S(buf, 1, t, row, kq)) = qb[t];
        }
    };
    bf16x8 fa[3][4], fb[3][4];
    auto fread = [&](int buf, int op, int t) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (op == 0) fa[t][i] = *reinterpret_cast<const bf16x8*>(S(buf, 0, t, wm * 128 + i * 32 + r, kh * 8));
            else fb[t][i] = *reinterpret_cast<const bf16x8*>(S(buf, 1, t, wn * 128 + i * 32 + r, kh * 8));
        }
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    auto group = [&](int ta, int tb) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ta][i], fb[tb][j], acc[i][j], 0, 0, 0);
    };
    const int nchunk = K / BK2;
    gload(0);
#pragma unroll
    for (int p = 0; p < 4; ++p) lstore(0, p);
    if (nchunk > 1) gload(BK2);
    __syncthreads();
    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nchunk;
        fread(cur, 0, 0); fread(cur, 1, 0); fread(cur, 1, 1); fread(cur, 0, 1); fread(cur, 1, 2); fread(cur, 0, 2);
        // 6 product groups x 4 row blocks = 24 slots of 4 MFMAs; the loader's work is cut into slices that ride behind the slots
        constexpr int GA[6] = {0, 0, 1, 1, 0, 2}, GB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
        for (int slot = 0; slot < 24; ++slot) {
            const int q = slot >> 2, i = slot & 3;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[GA[q]][i], fb[GB[q]][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (more && slot >= 2 && slot < 18 && ((slot - 2) & 3) == 0) lstore(cur ^ 1, (slot - 2) >> 2);     // slots 2, 6, 10, 14
            if (c + 2 < nchunk && slot == 19) gload((c + 2) * BK2);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int n = n0 + wn * 128 + j * 32 + r;
                C[(size_t)m * N + n] = acc[i][j][q];
            }
}

// the same tiling on the f32 MFMA (operands transposed into LDS as [k][row], one float per lane and k-step)
__global__ void __launch_bounds__(NT) k_gemm_f32(const float* __restrict__ A, const float* __restrict__ Bt, float* __restrict__ C,
                                                 int M, int N, int K) {
    constexpr int LDA = BM + 1;
    __shared__ float sA[BK][LDA];
    __shared__ float sB[BK][LDA];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int r = lane & 31, kh = lane >> 5;
    float4 ra[4], rb[4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * NT, row = idx >> 3, kq = (idx & 7) * 4;
            ra[p] = *reinterpret_cast<const float4*>(A + (size_t)(m0 + row) * K + k0 + kq);
            rb[p] = *reinterpret_cast<const float4*>(Bt + (size_t)(n0 + row) * K + k0 + kq);
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = tid + p * NT, row = idx >> 3, kq = (idx & 7) * 4;
            sA[kq][row] = ra[p].x; sA[kq + 1][row] = ra[p].y; sA[kq + 2][row] = ra[p].z; sA[kq + 3][row] = ra[p].w;
            sB[kq][row] = rb[p].x; sB[kq + 1][row] = rb[p].y; sB[kq + 2][row] = rb[p].z; sB[kq + 3][row] = rb[p].w;
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        lstore();
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = sA[2 * ks + kh][wm * 64 + i * 32 + r];
                b[i] = sB[2 * ks + kh][wn * 64 + i * 32 + r];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = m0 + wm * 64 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                const int n = n0 + wn * 64 + j * 32 + r;
                C[(size_t)m * N + n] = acc[i][j][q];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

static void run(int M, int N, int K) {
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K), hC((size_t)M * N);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f * 2.f - 1.f + ((s >> 24) & 0xff) * 1e-7f; };
    for (auto& v : hA) v = rnd();
    for (auto& v : hB) v = rnd();
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, hC.size() * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    dim3 grid(N / BN, M / BM), blk(NT);
    const double flops = 2.0 * M * N * K;
    auto check = [&](const char* name, double us) {
        CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, worst_scaled = 0;
        unsigned t = 777u;
        for (int q = 0; q < 400; ++q) {
            t = t * 1664525u + 1013904223u; const int m = (t >> 8) % M;
            t = t * 1664525u + 1013904223u; const int n = (t >> 8) % N;
            double ref = 0, mag = 0;
            for (int k = 0; k < K; ++k) {
                const double p = (double)hA[(size_t)m * K + k] * (double)hB[(size_t)n * K + k];
                ref += p; mag += std::fabs(p);
            }
            const double e = std::fabs((double)hC[(size_t)m * N + n] - ref);
            worst = std::fmax(worst, e / std::fmax(std::fabs(ref), 1e-30));
            worst_scaled = std::fmax(worst_scaled, e / mag);
        }
        printf("  %-34s %8.1f us  %7.1f TFLOP/s   max |err| / sum|a b| = %.2e   (max |err| / |c| = %.2e)\n", name, us, flops / us * 1e-6,
               worst_scaled, worst);
    };
    printf("M %d  N %d  K %d\n", M, N, K);
    double us = time_us([&]() { hipLaunchKernelGGL(k_gemm_f32, grid, blk, 0, 0, dA, dB, dC, M, N, K); }, 10);
    check("f32 MFMA 32x32x2", us);
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<6, false, false>), grid, blk, 0, 0, dA, dB, dC, M, N, K, nullptr, nullptr); }, 10);
    check("3 bf16 limbs, 6 products", us);
    us = time_us([&]() { hipLaunchKernelGGL(k_gemm_limb_pipe, grid, blk, 0, 0, dA, dB, dC, M, N, K); }, 10);
    check("3 limbs, 6 products, pipelined", us);
    us = time_us([&]() { hipLaunchKernelGGL(k_gemm_limb_pipe2, grid, blk, 0, 0, dA, dB, dC, M, N, K); }, 10);
    check("3 limbs, 6 products, pipelined 2", us);
    if (M % BM3 == 0 && N % BN3 == 0) {
        const size_t lds3 = (size_t)2 * 2 * 3 * 256 * LDK2 * 2;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_limb_big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
        us = time_us([&]() { hipLaunchKernelGGL(k_gemm_limb_big, dim3(N / BN3, M / BM3), blk, lds3, 0, dA, dB, dC, M, N, K); }, 10);
        check("3 limbs, 6 products, 128x128 per wave", us);
    }
    __bf16* dA3;
    CK(hipMalloc(&dA3, (size_t)3 * M * K * 2));
    hipLaunchKernelGGL(k_split, dim3(2048), dim3(256), 0, 0, dA, dA3, dA3 + (size_t)M * K, dA3 + (size_t)2 * M * K, (long)M * K);
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<6, true, false>), grid, blk, 0, 0, dA, dB, dC, M, N, K, dA3, nullptr); }, 10);
    check("3 limbs, 6 products, A pre-split", us);
    __bf16* dB3;
    CK(hipMalloc(&dB3, (size_t)3 * N * K * 2));
    hipLaunchKernelGGL(k_split, dim3(2048), dim3(256), 0, 0, dB, dB3, dB3 + (size_t)N * K, dB3 + (size_t)2 * N * K, (long)N * K);
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<6, true, true>), grid, blk, 0, 0, dA, dB, dC, M, N, K, dA3, dB3); }, 10);
    check("3 limbs, 6 products, A and B pre-split", us);
    us = time_us([&]() { hipLaunchKernelGGL(k_split, dim3(2048), dim3(256), 0, 0, dB, dB3, dB3 + (size_t)N * K, dB3 + (size_t)2 * N * K, (long)N * K); }, 10);
    printf("  %-34s %8.1f us  (the pass that splits B once: %.1f MB read, %.1f MB written)\n", "k_split(B)", us, N * (double)K * 4e-6, N * (double)K * 6e-6);
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<3, false, false>), grid, blk, 0, 0, dA, dB, dC, M, N, K, nullptr, nullptr); }, 10);
    check("2 bf16 limbs, 3 products", us);
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<3, true, false>), grid, blk, 0, 0, dA, dB, dC, M, N, K, dA3, nullptr); }, 10);
    check("2 limbs, 3 products, A pre-split", us);
    CK(hipFree(dA3)); CK(hipFree(dB3));
    us = time_us([&]() { hipLaunchKernelGGL((k_gemm_limb<1, false, false>), grid, blk, 0, 0, dA, dB, dC, M, N, K, nullptr, nullptr); }, 10);
    check("bf16 (1 product)", us);
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
}

int main() {
    run(4096, 4096, 4096);
    run(512, 46080, 1152);          // a ResNet layer3-like conv as a GEMM: Cout 512... x (12 x 24 x 160 pixels) x (128 channels x 9 taps)
    return 0;
}
