// Split-precision arithmetic shared by conv_limb.hip (the GEMM kernels) and conv.hip (the batched weight re-layout):
// an fp32 value is the exact sum of three bfloat16 "limbs" x = h + m + l (round-to-nearest at every level: 8 + 8 + 8 signed
// mantissa bits cover the 24 of an fp32), and a product a * b is formed on the bf16 matrix pipes as the six limb products
// ah*bh + ah*bm + am*bh + am*bm + ah*bl + al*bh with fp32 accumulation; the three dropped terms are <= 2^-26 |a b|, a quarter
// of the rounding error of one fp32 multiplication.  Every bf16 x bf16 product is exact in fp32, so the result carries the
// accuracy of an fp32 dot product (tests/test_gpu_limb.py: error against float64 next to the f32-MFMA kernel's).
#pragma once
#include <hip/hip_runtime.h>

namespace fdlimb {
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// two floats -> two bf16 (round to nearest even) in one register: v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    f32x2_t v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// a pair of floats -> the pair's three limb registers (13 vector instructions)
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m, unsigned& l) {
    h = pk_bf16(a, b);
    const float ra = a - bf_lo(h), rb = b - bf_hi(h);      // exact
    m = pk_bf16(ra, rb);
    l = pk_bf16(ra - bf_lo(m), rb - bf_hi(m));             // the second residual is exact too; l rounds nothing away
}
// eight consecutive-K floats -> the three 16-byte MFMA operand pieces
__device__ __forceinline__ void split8(const float (&x)[8], uint4& h, uint4& m, uint4& l) {
    split2(x[0], x[1], h.x, m.x, l.x);
    split2(x[2], x[3], h.y, m.y, l.y);
    split2(x[4], x[5], h.z, m.z, l.z);
    split2(x[6], x[7], h.w, m.w, l.w);
}

// Pre-split A operand (the weights; re-derived once per optimiser step): for GEMM row m and GEMM-K index kk the limb L lives at
//   A3[(((kk / 16) * 3 + L) * 2 + (kk / 8) % 2) * M + m][kk % 8]      (16-byte pieces of 8 bf16)
// i.e. one K-chunk of 16 holds, per limb and per half-chunk, the pieces of all M rows back to back - exactly the image the GEMM
// kernel wants in LDS (a lane's MFMA fragment = one piece; consecutive rows = consecutive 16-byte slots), so a tile's rows are
// one contiguous run per (chunk, limb, half).
__host__ __device__ inline long a3_piece(long kk, int L, long m, long M) {
    return ((((kk >> 4) * 3 + L) * 2 + ((kk >> 3) & 1)) * M + m);
}
__host__ __device__ inline long a3_floats(long M, long K) { return (3 * M * K + 1) / 2; }
}  // namespace fdlimb
