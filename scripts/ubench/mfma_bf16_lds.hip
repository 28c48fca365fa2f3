// Ceiling of an LDS-fed six-product limb step on gfx950: each wave repeatedly reads its 12 operand fragments (16 bytes per lane each)
// from LDS and issues the 24 v_mfma_f32_32x32x16_bf16 of a 64 x 64 x 16 step (3 limbs x 2 operands x 2 row blocks; six products per
// accumulator) - no global memory, no barriers, no limb splitting.  What the matrix pipes can reach when operands only come from LDS.
//   variants: FR = fragments re-read every step (like the GEMM), or read once (register-resident: the pure MFMA rate)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int LDK = 24;
template <bool REREAD, int NPROD>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[3][128][LDK];
    __shared__ __attribute__((aligned(16))) __bf16 sB[3][128][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, kh = lane >> 5;
    for (int i = tid; i < 3 * 128 * LDK; i += 256) { (&sA[0][0][0])[i] = (__bf16)(0.001f * (i % 7)); (&sB[0][0][0])[i] = (__bf16)(0.002f * (i % 5)); }
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    bf16x8 a[3][2], b[3][2];
    auto rd = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[t][i] = *reinterpret_cast<const volatile bf16x8*>(&sA[t][wm * 64 + i * 32 + r][kh * 8]);
                b[t][i] = *reinterpret_cast<const volatile bf16x8*>(&sB[t][wn * 64 + i * 32 + r][kh * 8]);
            }
    };
    rd();
    constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
    for (int it = 0; it < iters; ++it) {
        if (REREAD) rd();
#pragma unroll
        for (int q = 0; q < NPROD; ++q)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    out[blockIdx.x * 256 + tid] = s;
}
template <bool REREAD, int NPROD>
static void run(const char* name, int wgs_per_cu) {
    float* d; hipMalloc(&d, 256 * 256 * 8 * sizeof(float));
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<REREAD, NPROD>), dim3(grid), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<REREAD, NPROD>), dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_flops = (double)grid * 4 * iters * (4.0 * NPROD) * 32768.0;
    printf("%-52s %d workgroup(s) per CU: %7.1f us  %7.1f TFLOP/s of bf16 MFMA work = %.0f %% of 2.5 PFLOP/s\n", name, wgs_per_cu, ms * 1e3,
           mfma_flops / ms * 1e-9, 100.0 * mfma_flops / ms * 1e-9 / 2500.0);
    hipFree(d);
}
int main() {
    run<false, 6>("registers only, 6 products / 24 MFMAs per step", 1);
    run<false, 6>("registers only, 6 products / 24 MFMAs per step", 2);
    run<true, 6>("12 LDS fragments re-read per step, 24 MFMAs", 1);
    run<true, 6>("12 LDS fragments re-read per step, 24 MFMAs", 2);
    run<true, 6>("12 LDS fragments re-read per step, 24 MFMAs", 3);
    run<true, 1>("4 of the fragments used: 4 MFMAs per 12 reads", 2);
    return 0;
}
