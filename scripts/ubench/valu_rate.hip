// Micro-benchmark: sustained VALU issue rate on gfx950 for the instruction kinds the fused loss kernel is made of:
// v_fma_f32, v_pk_fma_f32, v_add_f32_dpp wave_shr, v_rcp_f32, v_cndmask.  Reports cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float x) {
    float a[8];
    f2 p[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f2{a[i], a[i] + 1.f}; }
    const f2 x2 = f2{x, x * 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], x, 0.5f);
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], x2, x2);
                if (MODE == 2) asm volatile("v_add_f32_dpp %0, %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if (MODE == 3) a[i] = __builtin_amdgcn_rcpf(a[i]);
                if (MODE == 4) a[i] = a[i] > x ? a[i] - 1.f : a[i] + x;
                if (MODE == 5) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
                if (MODE == 6) asm volatile("v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            }
        }
    }
    float v = 0.f;
    for (int i = 0; i < 8; ++i) v += a[i] + p[i][0] + p[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int MODE>
void run(const char* name, int blocks, int iters, double instr_per_elem) {
    float* out; hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 16 * 8 * instr_per_elem;
    printf("%-34s blocks %5d (%d waves/SIMD): %8.3f ms  %.3f T wave-instr/s  %.2f cycles/instr/SIMD @2.4GHz\n", name, blocks, blocks / 256, ms,
           winstr / (ms * 1e-3) / 1e12, (ms * 1e-3) * 2.4e9 / (winstr / 1024));
    hipFree(out);
}
int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("v_fma_f32", blocks, 2000, 1);
        run<1>("v_pk_fma_f32", blocks, 2000, 1);
        run<2>("v_add_f32_dpp wave_shr:1", blocks, 2000, 1);
        run<5>("v_add_f32_dpp row_shr:1", blocks, 2000, 1);
        run<6>("v_mov_b32_dpp wave_shr:1", blocks, 2000, 1);
        run<3>("v_rcp_f32", blocks, 2000, 1);
        run<4>("cmp+sub+add+cndmask (4 instr)", blocks, 2000, 4);
    }
    return 0;
}
