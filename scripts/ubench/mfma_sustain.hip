// Sustained fp32 MFMA rate under the board's power management: v_mfma_f32_32x32x2_f32 back to back on every SIMD of the chip for
// several seconds (random-mantissa operands; zeros would draw less power), reported per ~0.5 s window together with what the
// achieved rate says about the clock: rate / (256 CUs x 4 SIMDs x 64 flop/cycle) = effective MFMA-issue frequency.
//   DUTY d (argv[1], percent): every wave issues MFMAs for d % of its instruction slots and s_nop's for the rest - the matrix pipes'
//   busy fraction of a convolution kernel (50-60 %) instead of a GEMM's.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_sustain mfma_sustain.hip ; run next to `rocm-smi --showclocks --showpower` samples
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int IDLE>      // IDLE: s_nop groups between MFMAs (each s_nop 15 = 16 idle cycles)
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc[4];
    const float a = seed * (float)((threadIdx.x * 2654435761u >> 9) & 0xffff) * (1.0f / 65536.0f) + 0.37f;
    const float b = seed * (float)((threadIdx.x * 40503u + blockIdx.x * 977u) & 0xffff) * (1.0f / 65536.0f) - 0.41f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a + (float)u, b - (float)i, acc[i], 0, 0, 0);
#pragma unroll
                for (int z = 0; z < IDLE; ++z) asm volatile("s_nop 15");
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
template <int IDLE>
void run(const char* name, float* out, double seconds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000, blocks = 256 * 2;         // 2 workgroups of 4 waves per CU: 2 waves per SIMD
    const double flop = (double)blocks * 4 /*waves*/ * iters * 32 /*mfma*/ * 32.0 * 32 * 2 * 2;
    double total = 0;
    printf("%s\n", name);
    while (total < seconds) {
        hipEventRecord(e0);
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k<IDLE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total += ms * 1e-3;
        const double tf = 4 * flop / (ms * 1e-3) / 1e12;
        printf("  t = %5.2f s: %6.1f TFLOP/s  (MFMA issue clock %.0f MHz at 100 %% duty)\n", total, tf, tf * 1e12 / (256.0 * 4 * 64) / 1e6);
        fflush(stdout);
    }
}
int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 1 << 20);
    const int idle = argc > 1 ? atoi(argv[1]) : 0;
    const double secs = argc > 2 ? atof(argv[2]) : 8.0;
    if (idle == 0) run<0>("fp32 MFMA back to back (100 % duty)", out, secs);
    else if (idle == 2) run<2>("fp32 MFMA, 64 busy + 32 idle cycles (2 waves per SIMD cover each other: ~100 % pipe duty, lower issue pressure)", out, secs);
    else run<4>("fp32 MFMA, 64 busy + 64 idle cycles per wave", out, secs);
    return 0;
}
