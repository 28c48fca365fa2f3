// Micro-benchmark 3 (round 5): what would a PACKED two-columns-per-lane formulation of the loss kernel buy on gfx950?
// Rates of v_pk_{fma,mul,add}_f32 (two fp32 operations per lane and instruction, 64-bit register pairs) against the plain forms,
// and of the DPP wave shifts the 3x3 window sums are made of.  8 independent accumulators per lane, inline asm so that the operand
// form is exactly what is named.  Reports cycles per wave-instruction per SIMD at the 2.4 GHz nominal clock, and - the number that
// matters - cycles per 64 fp32 LANE-OPERATIONS (a packed instruction does 128 of them).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate3 valu_rate3.hip && ./valu_rate3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));
#define OP8(STR) \
    asm volatile(STR : "+v"(a[0]) : "v"(b0), "v"(b1)); asm volatile(STR : "+v"(a[1]) : "v"(b1), "v"(b2)); \
    asm volatile(STR : "+v"(a[2]) : "v"(b2), "v"(b3)); asm volatile(STR : "+v"(a[3]) : "v"(b3), "v"(b0)); \
    asm volatile(STR : "+v"(a[4]) : "v"(b0), "v"(b2)); asm volatile(STR : "+v"(a[5]) : "v"(b1), "v"(b3)); \
    asm volatile(STR : "+v"(a[6]) : "v"(b2), "v"(b0)); asm volatile(STR : "+v"(a[7]) : "v"(b3), "v"(b1));
#define OP8S(STR) \
    asm volatile(STR : "+v"(s[0]) : "v"(c0), "v"(c1)); asm volatile(STR : "+v"(s[1]) : "v"(c1), "v"(c2)); \
    asm volatile(STR : "+v"(s[2]) : "v"(c2), "v"(c3)); asm volatile(STR : "+v"(s[3]) : "v"(c3), "v"(c0)); \
    asm volatile(STR : "+v"(s[4]) : "v"(c0), "v"(c2)); asm volatile(STR : "+v"(s[5]) : "v"(c1), "v"(c3)); \
    asm volatile(STR : "+v"(s[6]) : "v"(c2), "v"(c0)); asm volatile(STR : "+v"(s[7]) : "v"(c3), "v"(c1));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
    float2_ a[8];
    for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 0.001f + i; a[i].y = a[i].x * 0.5f; }
    float2_ b0 = a[0] * 0.5f, b1 = a[1] * 0.25f, b2 = a[2] * 0.125f, b3 = a[3] * 0.3f;
    float s[8];
    for (int i = 0; i < 8; ++i) s[i] = a[i].y;
    float c0 = b0.x, c1 = b1.x, c2 = b2.x, c3 = b3.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) { OP8("v_pk_fma_f32 %0, %0, %1, %2") }
            if (MODE == 1) { OP8("v_pk_mul_f32 %0, %0, %1") }
            if (MODE == 2) { OP8("v_pk_add_f32 %0, %0, %1") }
            if (MODE == 3) { OP8S("v_fma_f32 %0, %0, %1, %2") }       // plain forms on the low halves, for the same-run yardstick
            if (MODE == 4) { OP8S("v_add_f32_e32 %0, %0, %1") }
            if (MODE == 5) { OP8S("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") }
            if (MODE == 6) { OP8S("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") }
            if (MODE == 7) { OP8S("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1") }
            if (MODE == 8) { OP8("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,1]") }   // broadcast of a scalar-like low half
            if (MODE == 9) { OP8("v_pk_mov_b32 %0, %1, %2") }
        }
    }
    float v = 0.f;
    for (int i = 0; i < 8; ++i) v += a[i].x + a[i].y + s[i];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int MODE>
void run(const char* name, int lane_ops, int blocks, int iters) {
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 16 * 8;
    double cyc = (ms * 1e-3) * 2.4e9 / (winstr / 1024);
    printf("%-58s %d waves/SIMD: %.2f cycles/instr/SIMD = %.2f cycles per 64 fp32 lane-operations\n", name, blocks / 256, cyc, cyc / lane_ops);
    (void)hipFree(out);
}
int main() {
    for (int blocks : {256, 512}) {
        run<3>("v_fma_f32 v,v,v,v (plain, 64 lane-ops)", 1, blocks, 2000);
        run<4>("v_add_f32 v,v,v (plain)", 1, blocks, 2000);
        run<0>("v_pk_fma_f32 (128 lane-ops)", 2, blocks, 2000);
        run<1>("v_pk_mul_f32", 2, blocks, 2000);
        run<2>("v_pk_add_f32", 2, blocks, 2000);
        run<8>("v_pk_fma_f32 op_sel_hi:[1,0,1]", 2, blocks, 2000);
        run<9>("v_pk_mov_b32", 2, blocks, 2000);
        run<5>("v_add_f32_dpp wave_shr:1", 1, blocks, 2000);
        run<7>("v_add_f32_dpp row_shr:1", 1, blocks, 2000);
        run<6>("v_mov_b32_dpp wave_shr:1", 1, blocks, 2000);
    }
    return 0;
}
