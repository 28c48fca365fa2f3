// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access widths the loss kernel uses: streams 1 GiB (4x the Infinity
// Cache) once with 4-byte and once with 16-byte loads per lane.  Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace`.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_read4(const float* __restrict__ p, float* out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void k_read16(const float4* __restrict__ p, float* out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t n = (size_t)256 << 20;          // floats = 1 GiB
    float *p, *o;
    (void)hipMalloc(&p, n * 4); (void)hipMalloc(&o, 4);
    (void)hipMemset(p, 0, n * 4);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_read4, dim3(4096), dim3(256), 0, 0, p, o, n);
        hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const float4*)p, o, n / 4);
    }
    (void)hipDeviceSynchronize();
    printf("read %zu bytes per launch\n", n * 4);
    return 0;
}
