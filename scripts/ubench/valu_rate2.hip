// Micro-benchmark 2: which VALU forms are full rate on gfx950?  8 independent accumulators per lane, inline asm so that the
// operand form is exactly what is named.  Reports cycles per wave-instruction per SIMD (at the 2.4 GHz nominal clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OP8(STR) \
    asm volatile(STR : "+v"(a[0]) : "v"(b0), "v"(b1), "s"(sx)); asm volatile(STR : "+v"(a[1]) : "v"(b1), "v"(b2), "s"(sx)); \
    asm volatile(STR : "+v"(a[2]) : "v"(b2), "v"(b3), "s"(sx)); asm volatile(STR : "+v"(a[3]) : "v"(b3), "v"(b0), "s"(sx)); \
    asm volatile(STR : "+v"(a[4]) : "v"(b0), "v"(b2), "s"(sx)); asm volatile(STR : "+v"(a[5]) : "v"(b1), "v"(b3), "s"(sx)); \
    asm volatile(STR : "+v"(a[6]) : "v"(b2), "v"(b0), "s"(sx)); asm volatile(STR : "+v"(a[7]) : "v"(b3), "v"(b1), "s"(sx));
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, float x) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
    float b0 = a[0] * 0.5f, b1 = a[1] * 0.25f, b2 = a[2] * 0.125f, b3 = a[3] * 0.3f;
    const float sx = x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (MODE == 0) { OP8("v_add_f32_e32 %0, %0, %1") }
            if (MODE == 1) { OP8("v_mul_f32_e32 %0, %0, %1") }
            if (MODE == 2) { OP8("v_fma_f32 %0, %0, %1, %2") }
            if (MODE == 3) { OP8("v_fmac_f32_e32 %0, %1, %2") }
            if (MODE == 4) { OP8("v_fma_f32 %0, %0, %3, %1") }
            if (MODE == 5) { OP8("v_cndmask_b32_e32 %0, %0, %1, vcc") }
            if (MODE == 6) { OP8("v_max_f32_e32 %0, %0, %1") }
            if (MODE == 7) { OP8("v_sub_f32_e32 %0, %1, %0") }
            if (MODE == 8) { OP8("v_mul_f32_e64 %0, %0, %3") }
            if (MODE == 9) { OP8("v_fma_f32 %0, %1, %2, %0") }
            if (MODE == 10) { OP8("v_add_f32_e64 %0, %0, |%1|") }
            if (MODE == 11) { OP8("v_med3_f32 %0, %0, %1, %2") }
            if (MODE == 12) { OP8("v_mov_b32_e32 %0, %1") }
            if (MODE == 13) { OP8("v_add_u32_e32 %0, %0, %1") }
            if (MODE == 14) { OP8("v_mad_u32_u24 %0, %0, %1, %2") }
            if (MODE == 15) { OP8("v_cmp_lt_f32_e32 vcc, %0, %1") }
            if (MODE == 16) { OP8("v_mul_f32_e32 %0, 0.5, %0") }
            if (MODE == 17) { OP8("v_mul_f32_e32 %0, 0x40900000, %0") }
            if (MODE == 18) { OP8("v_fmaak_f32 %0, %0, %1, 0x40900000") }
            if (MODE == 19) { OP8("v_fmamk_f32 %0, %0, 0x3f800001, %1") }
            if (MODE == 20) { OP8("v_add_f32_e32 %0, 1.0, %0") }
            if (MODE == 21) { OP8("v_fma_f32 %0, %0, 2.0, %1") }
            if (MODE == 22) { OP8("v_min_f32_e32 %0, %0, %1") }
            if (MODE == 23) { OP8("v_floor_f32_e32 %0, %1") }
            if (MODE == 24) { OP8("v_cvt_i32_f32_e32 %0, %1") }
            if (MODE == 25) { OP8("v_lshlrev_b32_e32 %0, 2, %0") }
            if (MODE == 26) { OP8("v_fract_f32_e32 %0, %1") }
            if (MODE == 27) { OP8("v_mul_legacy_f32 %0, %0, %1") }
        }
    }
    float v = 0.f;
    for (int i = 0; i < 8; ++i) v += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int MODE>
void run(const char* name, int blocks, int iters) {
    float* out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * 4 * iters * 16 * 8;
    printf("%-36s %d waves/SIMD: %.2f cycles/instr/SIMD @2.4GHz\n", name, blocks / 256, (ms * 1e-3) * 2.4e9 / (winstr / 1024));
    (void)hipFree(out);
}
int main() {
    for (int blocks : {512}) {
        run<0>("v_add_f32_e32 v,v,v", blocks, 2000);
        run<1>("v_mul_f32_e32 v,v,v", blocks, 2000);
        run<7>("v_sub_f32_e32 v,v,v", blocks, 2000);
        run<6>("v_max_f32_e32 v,v,v", blocks, 2000);
        run<2>("v_fma_f32 a,a,b,c (3 vgpr)", blocks, 2000);
        run<9>("v_fma_f32 a,b,c,a (3 vgpr)", blocks, 2000);
        run<3>("v_fmac_f32_e32 a,b,c", blocks, 2000);
        run<4>("v_fma_f32 a,a,s,b (sgpr)", blocks, 2000);
        run<8>("v_mul_f32_e64 a,a,s", blocks, 2000);
        run<10>("v_add_f32_e64 a,a,|b|", blocks, 2000);
        run<11>("v_med3_f32 a,a,b,c", blocks, 2000);
        run<5>("v_cndmask_b32_e32 a,a,b,vcc", blocks, 2000);
        run<12>("v_mov_b32 a,b", blocks, 2000);
        run<13>("v_add_u32 a,a,b", blocks, 2000);
        run<14>("v_mad_u32_u24 a,a,b,c", blocks, 2000);
        run<15>("v_cmp_lt_f32 vcc,a,b", blocks, 2000);
        run<16>("v_mul_f32 a,0.5,a (inline)", blocks, 2000);
        run<17>("v_mul_f32 a,literal,a", blocks, 2000);
        run<18>("v_fmaak_f32 a,a,b,literal", blocks, 2000);
        run<19>("v_fmamk_f32 a,a,literal,b", blocks, 2000);
        run<20>("v_add_f32 a,1.0,a (inline)", blocks, 2000);
        run<21>("v_fma_f32 a,a,2.0,b (inline)", blocks, 2000);
        run<22>("v_min_f32 a,a,b", blocks, 2000);
        run<23>("v_floor_f32 a,b", blocks, 2000);
        run<24>("v_cvt_i32_f32 a,b", blocks, 2000);
        run<25>("v_lshlrev_b32 a,2,a", blocks, 2000);
        run<26>("v_fract_f32 a,b", blocks, 2000);
        run<27>("v_mul_legacy_f32 a,a,b", blocks, 2000);
    }
    return 0;
}
