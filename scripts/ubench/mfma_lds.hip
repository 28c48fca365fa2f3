// Micro-benchmark: cost of feeding v_mfma_f32_32x32x2_f32 (64x64 outputs per wave) from LDS:
//   V0 ds_read_b32 operands, read-then-use          V1 same, software-pipelined one k-step ahead
//   V2 ds_read_b128 operands ([m][k] layout, k-pairs (s, s+16)), 4 k-steps per read, pipelined
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int V>
__global__ void __launch_bounds__(256) k(float* out, int chunks) {
    extern __shared__ __attribute__((aligned(16))) float s[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 260; i += 256) s[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    for (int ch = 0; ch < chunks; ++ch) {
        const int cur = ch & 1;
        if (V < 2) {
            const float* pa = s + cur * 32 * 260 + (lane >> 5) * 129 + (wave >> 1) * 64 + (lane & 31);
            const float* pb = s + cur * 32 * 260 + 32 * 129 + (lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31);
            float av[2][2], bv[2][2];
            av[0][0] = pa[0]; av[0][1] = pa[32]; bv[0][0] = pb[0]; bv[0][1] = pb[32];
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int cb = V == 1 ? (kk & 1) : 0, nb = V == 1 ? (cb ^ 1) : 0;
                if (V == 1) { if (kk + 1 < 16) { av[nb][0] = pa[(kk + 1) * 258]; av[nb][1] = pa[(kk + 1) * 258 + 32]; bv[nb][0] = pb[(kk + 1) * 256]; bv[nb][1] = pb[(kk + 1) * 256 + 32]; } }
                else { av[0][0] = pa[kk * 258]; av[0][1] = pa[kk * 258 + 32]; bv[0][0] = pb[kk * 256]; bv[0][1] = pb[kk * 256 + 32]; }
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // [row][k] layout, row stride 36 floats; lane (i = lane&31, h = lane>>5) reads k = 16h + 4g .. +3
            const float* pa = s + cur * 32 * 260 + ((wave >> 1) * 64 + (lane & 31)) * 36 + (lane >> 5) * 16;
            const float* pb = s + cur * 32 * 260 + 128 * 36 / 2 + ((wave & 1) * 64 + (lane & 31)) * 36 + (lane >> 5) * 16;
            float4 a0[2], a1[2], b0[2], b1[2];
            a0[0] = *(const float4*)(pa); a1[0] = *(const float4*)(pa + 32 * 36); b0[0] = *(const float4*)(pb); b1[0] = *(const float4*)(pb + 32 * 36);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cb = g & 1, nb = cb ^ 1;
                if (g + 1 < 4) { a0[nb] = *(const float4*)(pa + 4 * (g + 1)); a1[nb] = *(const float4*)(pa + 32 * 36 + 4 * (g + 1));
                                 b0[nb] = *(const float4*)(pb + 4 * (g + 1)); b1[nb] = *(const float4*)(pb + 32 * 36 + 4 * (g + 1)); }
                __builtin_amdgcn_sched_barrier(0);
                const float A0[4] = {a0[cb].x, a0[cb].y, a0[cb].z, a0[cb].w}, A1[4] = {a1[cb].x, a1[cb].y, a1[cb].z, a1[cb].w};
                const float B0[4] = {b0[cb].x, b0[cb].y, b0[cb].z, b0[cb].w}, B1[4] = {b1[cb].x, b1[cb].y, b1[cb].z, b1[cb].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[q], B0[q], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[q], B1[q], acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[q], B0[q], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[q], B1[q], acc[3], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float v = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) v += acc[a][r];
    out[blockIdx.x * 256 + tid] = v;
}
template <int V>
void run(const char* name, int blocks, int chunks, float* out) {
    const size_t lds = sizeof(float) * 2 * 32 * 260;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), lds, 0, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V>), dim3(blocks), dim3(256), lds, 0, out, chunks);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mfma = (double)blocks * 4 * chunks * 64;
    printf("%-46s blocks %4d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out; hipMalloc(&out, 4 << 20);
    for (int blocks : {256, 512}) {
        run<0>("b32 operands, read-then-use", blocks, 400, out);
        run<1>("b32 operands, pipelined 1 step ahead", blocks, 400, out);
        run<2>("b128 operands ([row][k]), pipelined", blocks, 400, out);
    }
    return 0;
}
