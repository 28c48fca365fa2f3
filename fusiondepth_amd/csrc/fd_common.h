// Shared device/host helpers for libfdhip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>

#define FD_WAVE 64

// ---- tuning (include/fdhip.h: fd_tuning; csrc/tuning.hip) - the library never reads the environment
struct fd_tuning;
const fd_tuning& fd_tun();

// ---- error plumbing (C ABI never throws; see include/fdhip.h) ------------------------------------
void fd_set_error(const char* fmt, ...);

#define FD_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            fd_set_error(__VA_ARGS__);        \
            return -1;                        \
        }                                     \
    } while (0)

#define FD_LAUNCH_CHECK(name)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            fd_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
            return (int)e__;                                                               \
        }                                                                                  \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize (> 64 KiB of dynamic LDS) is a PER-DEVICE property of a kernel: a launcher keeps one
// of these per kernel family - bit d says that device d has the attribute.  Two host threads may both set it (idempotent); the bit
// is published after the attribute is in place, so a thread that sees it never launches without.
struct FdLdsAttrOnce {
    std::atomic<unsigned long long> done{0};
    static int dev_bit() { int dev = 0; (void)hipGetDevice(&dev); return dev & 63; }
    bool needed() const { return !((done.load(std::memory_order_acquire) >> dev_bit()) & 1ull); }
    void mark() { done.fetch_or(1ull << dev_bit(), std::memory_order_release); }
};

static inline int fd_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- reductions ----------------------------------------------------------------------------------
// Deterministic: fixed shuffle tree inside a wave, fixed order across waves, no float atomics.
__device__ __forceinline__ float fd_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, FD_WAVE);
    return v;  // valid in lane 0
}

__device__ __forceinline__ float fd_wave_sum_all(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, FD_WAVE);
    return v;  // valid in every lane
}

// Sum of N values per thread over a block of NWAVES*64 threads.  `red` is LDS scratch of
// NWAVES*N floats.  Result valid in threads tid < N (thread i holds sum of value i).
template <int N, int NWAVES>
__device__ __forceinline__ float fd_block_sum_n(const float (&v)[N], float* red) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float s = fd_wave_sum(v[i]);
        if (lane == 0) red[wv * N + i] = s;
    }
    __syncthreads();
    float out = 0.f;
    if (tid < N) {
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) out += red[w * N + tid];
    }
    __syncthreads();
    return out;
}

// ---- index helpers ------------------------------------------------------------------------------
__device__ __forceinline__ int fd_reflect(int i, int n) {  // ReflectionPad2d(1)-style, valid for -n < i < 2n-1
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ int fd_clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// PyTorch upsample_bilinear2d(align_corners=False) source index (aten UpSample.h:
// area_pixel_compute_source_index): src = scale*(dst+0.5)-0.5 clamped at 0; i1 = min(i0+1, n-1).
__device__ __forceinline__ void fd_bilinear_src(int dst, float scale, int n_in, int& i0, int& i1, float& l1) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = (int)s;
    if (i0 > n_in - 1) i0 = n_in - 1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = s - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
}

// Raw buffer resources: 32-bit byte offsets against a wave-uniform base (one VALU add per load instead of a 64-bit
// address chain) and hardware range checking - an offset >= 2^31 reads as 0.0f without touching memory, which is how
// padding taps, pixels past the end and the "no next chunk" case are expressed (no selects, no branches in the loop).
constexpr unsigned FD_OOB = 0x80000000u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fd_make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)FD_OOB, 0x00020000);
}
__device__ __forceinline__ float fd_ldg32(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float4 fd_ldg128(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

#ifdef FD_SKIP_ABLATION      // timing experiments only: scripts/ubench/fd_skip.h (not part of the product build)
#include "fd_skip.h"
#endif
