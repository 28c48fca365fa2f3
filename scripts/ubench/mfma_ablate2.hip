// Ablation micro-benchmark, round 3: which non-MFMA instructions cost the fp32 MFMA loop its matrix-pipe time, and does their
// PLACEMENT matter?  One wave owns a 64x64 accumulator block (4 x v_mfma_f32_32x32x2_f32 per k-step, 8 k-steps per chunk, like
// k_conv_wino); per chunk and thread: N32 dword global loads, N128 dwordx4 global loads, NST ds_write_b32.
//   PLACE 0: memory ops ahead of the k-step's four MFMAs (sched_barrier on both sides) - what the product kernels do
//   PLACE 1: memory ops spread BETWEEN the four MFMAs of the k-step (one group after each MFMA)
//   PLACE 2: no sched_barrier at all (compiler's order)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NK = 8;
__device__ int g_rand = 0;      // 1: operands with random mantissas (power / clock behaviour of real data), 0: tiny constants
template <int N32, int N128, int NST, int PLACE, bool BAR>
__global__ void __launch_bounds__(256) k(const float* __restrict__ in, float* out, int chunks, unsigned stride) {
    extern __shared__ float s[];                      // 2 buffers x 16 x (65 + 64) x 4 comps ~ 66 KB like k_conv_wino, or less
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * 16 * 129; i += 256)
        s[i] = g_rand ? (float)(((unsigned)(i + 977 * blockIdx.x) * 2654435761u) >> 8 & 0xffffu) * (1.0f / 65536.0f) - 0.5f : (float)(i % 7) * 0.01f;
    __syncthreads();
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float rb[N32 > 0 ? N32 : 1]; float4 ra[N128 > 0 ? N128 : 1];
    unsigned off = (blockIdx.x * 256 + tid) % 4096;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 0x7fffffff, 0x00020000);
    for (int ch = 0; ch < chunks; ++ch) {
        const int cur = ch & 1;
        const float* pa = s + cur * 16 * 129 + (lane >> 5) * 65 + (lane & 31);
        const float* pb = s + cur * 16 * 129 + 16 * 65 + (lane >> 5) * 64 + (lane & 31);
        float* q = s + (cur ^ 1) * 16 * 129 + (tid & 63) + (wave & 3) * 129;
        float av[2][2], bv[2][2];
        av[0][0] = pa[0]; av[0][1] = pa[32]; bv[0][0] = pb[0]; bv[0][1] = pb[32];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int cb = kk & 1, nb = cb ^ 1;
            auto reads = [&]() __attribute__((always_inline)) {
                if (kk + 1 < NK) {
                    av[nb][0] = pa[(kk + 1) * 2 * 65]; av[nb][1] = pa[(kk + 1) * 2 * 65 + 32];
                    bv[nb][0] = pb[(kk + 1) * 2 * 64]; bv[nb][1] = pb[(kk + 1) * 2 * 64 + 32];
                }
            };
            // memory work of this k-step, split in 4 groups g = 0..3
            auto memops = [&](int g) __attribute__((always_inline)) {
                if (kk < NK / 2) {          // loads in the first half of the chunk
#pragma unroll
                    for (int i = 0; i < N32; ++i)
                        if ((i * (NK / 2)) / (N32 > 0 ? N32 : 1) == kk && (i & 3) == g)
                            rb[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(4u * (off + i * stride)), 0, 0));
#pragma unroll
                    for (int i = 0; i < N128; ++i)
                        if ((i * (NK / 2)) / (N128 > 0 ? N128 : 1) == kk && (i & 3) == g)
                            ra[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(4u * ((off + i * 1024u) & ~3u)), 0, 0));
                } else {                    // LDS stores in the second half
#pragma unroll
                    for (int i = 0; i < NST; ++i)
                        if ((i * (NK / 2)) / (NST > 0 ? NST : 1) == kk - NK / 2 && (i & 3) == g) {
                            float v = (N32 > 0) ? rb[i % (N32 > 0 ? N32 : 1)] : ((N128 > 0) ? ra[i % (N128 > 0 ? N128 : 1)].x : av[cb][0]);
                            q[(i % 16) * 129 * 0 + (i * 67) % 1800] = v;
                        }
                }
            };
            if (PLACE == 0) {
                reads(); memops(0); memops(1); memops(2); memops(3);
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            } else if (PLACE == 1) {
                __builtin_amdgcn_sched_barrier(0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                reads(); memops(0);
                __builtin_amdgcn_sched_barrier(0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][1], acc[1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                memops(1);
                __builtin_amdgcn_sched_barrier(0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][0], acc[2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                memops(2);
                __builtin_amdgcn_sched_barrier(0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                memops(3);
            } else {
                reads(); memops(0); memops(1); memops(2); memops(3);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][0], bv[cb][1], acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][0], acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cb][1], bv[cb][1], acc[3], 0, 0, 0);
            }
        }
        // every loaded value is consumed once per chunk (otherwise the loads are dead code): costs one s_waitcnt like the
        // product kernels' LDS stores
#pragma unroll
        for (int i = 0; i < N32; ++i) asm volatile("" :: "v"(rb[i]));
#pragma unroll
        for (int i = 0; i < N128; ++i) asm volatile("" :: "v"(ra[i].x), "v"(ra[i].w));
        if (BAR) __syncthreads();
        off = (off + 37u) % 4096;
    }
    float v = 0.f;
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) v += acc[a][r];
    out[blockIdx.x * 256 + tid] = v;
}
template <int N32, int N128, int NST, int PLACE, bool BAR>
void run(const char* name, int blocks, int lds_kb, const float* in, float* out) {
    const int chunks = 800;
    const size_t lds = (size_t)lds_kb * 1024;
    auto kern = k<N32, N128, NST, PLACE, BAR>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, in, out, 4, 7680u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, in, out, chunks, 7680u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double mfma = (double)blocks * 4 * chunks * NK * 4;
    printf("%-58s blocks %4d lds %3d KB: %7.3f ms  %6.1f TFLOP/s\n", name, blocks, lds_kb, ms, mfma * 4096 / (ms * 1e-3) / 1e12);
}
int main(int argc, char** argv) {
    float *in, *out; hipMalloc(&in, 64 << 20); hipMalloc(&out, 4 << 20); hipMemset(in, 0, 64 << 20);
    const int rnd = argc > 1 && argv[1][0] == 'r';
    if (rnd) {
        float* h = (float*)malloc(64 << 20);
        unsigned x = 12345u;
        for (size_t i = 0; i < (64u << 20) / 4; ++i) { x = x * 1664525u + 1013904223u; h[i] = (float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f; }
        hipMemcpy(in, h, 64 << 20, hipMemcpyHostToDevice);
        hipMemcpyToSymbol(HIP_SYMBOL(g_rand), &rnd, sizeof(int));
        printf("== random operands ==\n");
    }
    struct Cfg { int blocks, lds; } cfgs[] = {{512, 66}, {768, 50}};
    for (auto c : cfgs) {
        run<0, 0, 0, 0, false>("MFMA + LDS reads", c.blocks, c.lds, in, out);
        run<0, 0, 0, 0, true>("+ barrier", c.blocks, c.lds, in, out);
        run<0, 0, 32, 0, true>("+ barrier + 32 ds_write", c.blocks, c.lds, in, out);
        run<12, 4, 32, 0, true>("wino-like: 12 b32 + 4 b128 loads, 32 ds_write, PLACE 0", c.blocks, c.lds, in, out);
        run<12, 4, 32, 1, true>("wino-like, PLACE 1 (between MFMAs)", c.blocks, c.lds, in, out);
        run<12, 4, 32, 2, true>("wino-like, PLACE 2 (compiler)", c.blocks, c.lds, in, out);
        run<0, 7, 12, 0, true>("7 b128 loads, 12 ds_write, PLACE 0", c.blocks, c.lds, in, out);
        run<0, 7, 12, 1, true>("7 b128 loads, 12 ds_write, PLACE 1", c.blocks, c.lds, in, out);
        run<16, 0, 0, 0, true>("16 b32 loads only, PLACE 0", c.blocks, c.lds, in, out);
        run<16, 0, 0, 1, true>("16 b32 loads only, PLACE 1", c.blocks, c.lds, in, out);
        run<0, 4, 0, 0, true>("4 b128 loads only, PLACE 0", c.blocks, c.lds, in, out);
        run<0, 16, 0, 0, true>("16 b128 loads only, PLACE 0", c.blocks, c.lds, in, out);
    }
    return 0;
}
