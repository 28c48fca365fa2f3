// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate (no memory traffic) and with LDS operand reads.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDS>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    __shared__ float s[32 * 260];
    for (int i = threadIdx.x; i < 32 * 260; i += 256) s[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int lane = threadIdx.x & 63;
    float a0 = lane * 0.001f, b0 = lane * 0.002f;
    const float* p = s + (lane >> 5) * 129 + (lane & 31);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av0 = a0, av1 = a0 + 1.f, bv0 = b0, bv1 = b0 + 1.f;
            if (LDS) { av0 = p[kk * 258]; av1 = p[kk * 258 + 32]; bv0 = p[kk * 258 + 64]; bv1 = p[kk * 258 + 96]; }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0], 0, 0, 0);
            if (NACC > 1) acc[1 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[1 % NACC], 0, 0, 0);
            if (NACC > 2) acc[2 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[2 % NACC], 0, 0, 0);
            if (NACC > 3) acc[3 % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[3 % NACC], 0, 0, 0);
        }
    }
    float v = 0.f;
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) v += acc[a][r];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int NACC, bool LDS>
void run(const char* name, int blocks, int iters) {
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mfma = (double)blocks * 4 * iters * 16 * NACC;
    printf("%-28s blocks %5d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz if 1024 SIMDs busy)\n", name, blocks, ms,
           mfma * 4096 / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (mfma / 1024));
    hipFree(out);
}
int main() {
    run<1, false>("1 acc, regs only, 1 blk/CU", 256, 2000);
    run<2, false>("2 acc, regs only, 1 blk/CU", 256, 2000);
    run<2, false>("2 acc, regs only, 3 blk/CU", 768, 1000);
    run<2, true>("2 acc, LDS operands, 1 blk/CU", 256, 2000);
    run<2, true>("2 acc, LDS operands, 3 blk/CU", 768, 1000);
    run<4, false>("4 acc, regs only", 256, 2000);
    run<4, false>("4 acc, regs only", 512, 2000);
    run<4, false>("4 acc, regs only", 1024, 1000);
    run<2, false>("2 acc, regs only", 512, 2000);
    run<4, true>("4 acc, LDS operands", 256, 2000);
    run<4, true>("4 acc, LDS operands", 512, 2000);
    run<4, true>("4 acc, LDS operands", 768, 1000);
    return 0;
}
