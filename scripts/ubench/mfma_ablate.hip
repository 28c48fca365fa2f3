// Ablation micro-benchmark: what stops a 64x64-per-wave fp32 MFMA loop from reaching peak?
// variants: +barrier per chunk, +LDS stores, +global loads (interleaved like k_conv_fast), 1 or 2 blocks per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <bool BAR, bool ST, bool LD>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* out, int chunks, unsigned stride) {
    extern __shared__ float s[];                      // 2 buffers x 32 x (129 + 128)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 32 * 257; i += 256) s[i] = (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float rb[16]; float4 ra[4];
    unsigned off = (blockIdx.x * 256 + tid) % 4096;
    for (int ch = 0; ch < chunks; ++ch) {
        const int cur = ch & 1;
        const float* pa = s + cur * 32 * 257 + (lane >> 5) * 129 + (wave >> 1) * 64 + (lane & 31);
        const float* pb = s + cur * 32 * 257 + 32 * 129 + (lane >> 5) * 128 + (wave & 1) * 64 + (lane & 31);
        float* qa = s + (cur ^ 1) * 32 * 257 + (4 * (tid % 8)) * 129 + tid / 8;
        float* qb = s + (cur ^ 1) * 32 * 257 + 32 * 129 + (tid / 128) * 128 + (tid % 128);
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float av0 = pa[kk * 2 * 129], av1 = pa[kk * 2 * 129 + 32], bv0 = pb[kk * 2 * 128], bv1 = pb[kk * 2 * 128 + 32];
            if (LD && kk < 8) {
                if ((kk & 1) == 0) ra[kk / 2] = *reinterpret_cast<const float4*>(in + ((off + kk * 1024u) & ~3u));
                rb[2 * kk] = in[off + (2 * kk) * stride];
                rb[2 * kk + 1] = in[off + (2 * kk + 1) * stride];
            }
            if (ST && kk >= 8) {
                const int j = kk - 8;
                if ((j & 1) == 0) { float4 v = LD ? ra[j / 2] : make_float4(av0, av1, bv0, bv1);
                    float* q = qa + (j / 2) * 32; q[0] = v.x; q[129] = v.y; q[258] = v.z; q[387] = v.w; }
                qb[(2 * j) * 2 * 128] = LD ? rb[2 * j] : av0;
                qb[(2 * j + 1) * 2 * 128] = LD ? rb[2 * j + 1] : bv0;
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv1, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (BAR) __syncthreads();
        off = (off + 37u) % 4096;
    }
    float v = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) v += acc[a][r];
    out[blockIdx.x * 256 + tid] = v;
}
template <bool BAR, bool ST, bool LD>
void run(const char* name, int blocks, int chunks, const float* in, float* out, unsigned stride) {
    const size_t lds = sizeof(float) * 2 * 32 * 257;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<BAR, ST, LD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<BAR, ST, LD>), dim3(blocks), dim3(256), lds, 0, in, out, 4, stride);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<BAR, ST, LD>), dim3(blocks), dim3(256), lds, 0, in, out, chunks, stride);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mfma = (double)blocks * 4 * chunks * 64;
    printf("%-44s blocks %4d: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}
int main() {
    float *in, *out; hipMalloc(&in, 64 << 20); hipMalloc(&out, 4 << 20); hipMemset(in, 0, 64 << 20);
    for (int blocks : {256, 512}) {
        run<false, false, false>("MFMA + LDS reads", blocks, 400, in, out, 7680);
        run<true, false, false>("+ barrier/chunk", blocks, 400, in, out, 7680);
        run<true, true, false>("+ barrier + LDS stores", blocks, 400, in, out, 7680);
        run<true, true, true>("+ barrier + LDS stores + global loads", blocks, 400, in, out, 7680);
        run<false, false, true>("MFMA + global loads only", blocks, 400, in, out, 7680);
        run<false, true, false>("MFMA + LDS stores only", blocks, 400, in, out, 7680);
    }
    return 0;
}
