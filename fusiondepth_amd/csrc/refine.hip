// Refine-decoder input construction (refiner.py:316-348) and the masked lower median it is built on ("fd_refine_inputs",
// "fd_masked_median", include/fdhip.h).
//
// Reference, per scale s:  disp (or, --refine_a0 true, the s-fold 2x2 ceil-mode max-pool of disp_0) -> bilinear upsample to
// H x W -> disp_to_depth -> depth *= median(beam[mask] * 100) / median(depth[mask]) (mask = LiDAR returns inside the crop
// rows 78..189 / columns 23..616, medians over the whole batch) -> scaled_disp = (bilinear down-sample of 1 / depth - 0.01) / 9.9,
// s-fold max-pool of depth -> Cat_xy, s-fold max-pool of the 2-channel map, concatenated.  ~60 ATen launches per step
// incl. five sorts (torch.median of a boolean selection) in rounds 2-4; here four launches:
//   k_refine_compact   the selected pixel indices of the batch, once (the mask is the same for all five medians)
//   k_refine_pool      --refine_a0: max over the r x r blocks of disp_0 for r = 2, 4, 8 (== repeated 2x2 ceil-mode pooling)
//   k_refine_medians   one workgroup per scale: radix select (4 passes of 8 bits over order-preserving integer keys, LDS
//                      histograms with integer atomics: no sort, order-independent => deterministic) of the LiDAR values and of
//                      the depths at the selected pixels - evaluated on the fly from the low-resolution disparity; -> ratio[s]
//   k_refine_outputs   every output element of every scale straight into the concatenated tensor: each full-resolution depth
//                      it needs is recomputed from <= 4 disparity taps (max-pooling commutes with the monotone disp -> depth map)
// HBM-bound by construction: reads disp_s (+ disp_0), the 2-channel map once, writes the outputs once.
#include "../../include/fdhip.h"
#include "fd_common.h"

namespace {

// torch.median semantics: the LOWER median = element of rank (n - 1) / 2 of the sorted selection; NaN if any element is NaN.
__device__ __forceinline__ unsigned key_of(float v) {     // order-preserving: a < b  <=>  key(a) < key(b)  (-0 < +0, NaNs last / first)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float value_of(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __builtin_bit_cast(float, u);
}

struct RefArgs {
    fd_refine_cfg cfg;
    float lo, span;
    const float* disp[4];          // the disparity the scale starts from, [B,1,Hs,Ws] (pooled disp_0 when cfg.pool_disp0)
    const float* beam; const float* two_cha;
    const float* inv_K[4];
    float* out[4];
    int* idx; int* count;          // compacted selection: flat pixel index b * H * W + y * W + x
    unsigned* keys;                // [n_scales][cap] scratch of the select passes
    float* ratio;                  // [n_scales][4]: ratio, median(depth[mask]), median(beam[mask] * 100), n
    int cap;
    int first_block[5];
};

__global__ void __launch_bounds__(256) k_refine_compact(const float* __restrict__ beam, int B, int H, int W, int y0, int y1, int x0,
                                                         int x1, int* __restrict__ idx, int* __restrict__ count) {
    const int cw = x1 - x0, ch = y1 - y0;
    const long n = (long)B * ch * cw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / ((long)ch * cw));
        const int r = (int)(i - (long)b * ch * cw);
        const int y = y0 + r / cw, x = x0 + r % cw;
        const int p = (b * H + y) * W + x;
        if (beam[p] > 0.f) idx[atomicAdd(count, 1)] = p;          // integer atomic: the ORDER of the list varies, its content does not
    }
}

// max over the r x r blocks of x (clipped at the border) == r-fold... log2(r)-fold F.max_pool2d(x, 2, ceil_mode=True)
__device__ __forceinline__ float block_max(const float* __restrict__ plane, int H, int W, int y, int x, int r) {
    float m = -INFINITY;
    const int ye = min(r * y + r, H), xe = min(r * x + r, W);
    for (int yy = r * y; yy < ye; ++yy)
        for (int xx = r * x; xx < xe; ++xx) m = fmaxf(m, plane[yy * W + xx]);
    return m;
}

struct PoolArgs { const float* x; float* out[3]; int planes, H, W; int Hs[3], Ws[3]; int first[4]; };
__global__ void __launch_bounds__(256) k_refine_pool(PoolArgs a) {
    int l = 0;
    while (l < 2 && (int)blockIdx.x >= a.first[l + 1]) ++l;
    const int r = 2 << l, Hs = a.Hs[l], Ws = a.Ws[l];
    const long n = (long)a.planes * Hs * Ws;
    for (long i = (long)(blockIdx.x - a.first[l]) * 256 + threadIdx.x; i < n; i += (long)(a.first[l + 1] - a.first[l]) * 256) {
        const int pl = (int)(i / ((long)Hs * Ws));
        const int q = (int)(i - (long)pl * Hs * Ws);
        a.out[l][i] = block_max(a.x + (long)pl * a.H * a.W, a.H, a.W, q / Ws, q % Ws, r);
    }
}

// bilinear upsampling of the scale's disparity at full-resolution pixel (y, x) (k_bilinear_fwd's arithmetic) -> depth (k_d2d_fwd's)
__device__ __forceinline__ float depth_at(const float* __restrict__ d, int Hs, int Ws, int H, int W, int y, int x, float lo, float span) {
    float dup;
    if (Hs == H && Ws == W) {
        dup = d[y * W + x];
    } else {
        int y0, y1, x0, x1;
        float ly, lx;
        fd_bilinear_src(y, (float)Hs / (float)H, Hs, y0, y1, ly);
        fd_bilinear_src(x, (float)Ws / (float)W, Ws, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        dup = hy * (hx * d[y0 * Ws + x0] + lx * d[y0 * Ws + x1]) + ly * (hx * d[y1 * Ws + x0] + lx * d[y1 * Ws + x1]);
    }
    const float s = lo + span * dup;
    return 1.0f / s;
}

// Radix select of the element of rank k among keys[0 .. n): 4 passes of 8 bits.  One workgroup of 1024 threads.
__device__ unsigned select_rank(const unsigned* __restrict__ keys, int n, int k, unsigned* hist, unsigned* sh) {
    unsigned prefix = 0, mask = 0;
    const int t = threadIdx.x;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (t < 256) hist[t] = 0;
        __syncthreads();
        for (int i = t; i < n; i += 1024) {
            const unsigned key = keys[i];
            if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (t == 0) {
            int left = k;
            unsigned bin = 0;
            for (; bin < 255; ++bin) {
                if (left < (int)hist[bin]) break;
                left -= (int)hist[bin];
            }
            sh[0] = bin; sh[1] = (unsigned)left;
        }
        __syncthreads();
        prefix |= sh[0] << shift;
        mask |= 255u << shift;
        k = (int)sh[1];
        __syncthreads();
    }
    return prefix;
}

// One workgroup per scale.  MODE of the launch: medians of beam * 100 and of depth_s over the compacted selection -> ratio[s].
__global__ void __launch_bounds__(1024) k_refine_medians(RefArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[4];
    __shared__ int nan_seen;
    const int s = blockIdx.x, t = threadIdx.x;
    const fd_refine_cfg& c = a.cfg;
    const int H = c.H, W = c.W, P = H * W;
    const int n = min(*a.count, a.cap);
    unsigned* keys = a.keys + (long)s * a.cap;
    float* r = a.ratio + s * 4;
    if (n == 0) {                      // torch.median of an empty selection raises in the reference; here: NaN, like the SI-log mean
        if (t == 0) { r[0] = r[1] = r[2] = __builtin_nanf(""); r[3] = 0.f; }
        return;
    }
    const int rank = (n - 1) / 2;
    float med[2];
    for (int which = 0; which < 2; ++which) {
        if (t == 0) nan_seen = 0;
        __syncthreads();
        for (int i = t; i < n; i += 1024) {
            const int p = a.idx[i];
            float v;
            if (which == 0) {
                v = a.beam[p] * 100.0f;
            } else {
                const int b = p / P, q = p - b * P;
                v = depth_at(a.disp[s] + (long)b * c.Hs[s] * c.Ws[s], c.Hs[s], c.Ws[s], H, W, q / W, q % W, a.lo, a.span);
            }
            if (v != v) nan_seen = 1;
            keys[i] = key_of(v);
        }
        __syncthreads();
        const unsigned key = select_rank(keys, n, rank, hist, sh);
        med[which] = nan_seen ? __builtin_nanf("") : value_of(key);
        __syncthreads();
    }
    if (t == 0) { r[0] = med[0] / med[1]; r[1] = med[1]; r[2] = med[0]; r[3] = (float)n; }
}

// Every element of out[s]: [B][1 + 3 catxy + 2][Hs][Ws] = (scaled_disp | x / 30, y / 2, (z - 40) / 40 | two_cha pooled)
__global__ void __launch_bounds__(256) k_refine_outputs(RefArgs a) {
    const fd_refine_cfg& c = a.cfg;
    int s = 0;
    while (s + 1 < c.n_scales && (int)blockIdx.x >= a.first_block[s + 1]) ++s;
    const int H = c.H, W = c.W, Hs = c.Hs[s], Ws = c.Ws[s];
    const int C = 1 + (c.catxy ? 3 : 0) + 2;
    const int rr = H / Hs;                                        // checked by the host: H == rr * Hs, W == rr * Ws
    const float ratio = a.ratio[s * 4];
    const long Ps = (long)Hs * Ws;
    const long n = (long)c.B * Ps;
    const int nblk = a.first_block[s + 1] - a.first_block[s];
    for (long i = (long)(blockIdx.x - a.first_block[s]) * 256 + threadIdx.x; i < n; i += (long)nblk * 256) {
        const int b = (int)(i / Ps);
        const int q = (int)(i - (long)b * Ps);
        const int y = q / Ws, x = q % Ws;
        const float* d = a.disp[s] + (long)b * Ps;
        float* o = a.out[s] + (long)b * C * Ps + q;
        // scaled_disp: F.interpolate(1 / depth, [Hs, Ws], bilinear, align_corners=False) of the full-resolution map (refiner.py:337-338)
        int y0, y1, x0, x1;
        float ly, lx;
        fd_bilinear_src(y, (float)H / (float)Hs, H, y0, y1, ly);
        fd_bilinear_src(x, (float)W / (float)Ws, W, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float i00 = 1.0f / (depth_at(d, Hs, Ws, H, W, y0, x0, a.lo, a.span) * ratio);
        const float i01 = 1.0f / (depth_at(d, Hs, Ws, H, W, y0, x1, a.lo, a.span) * ratio);
        const float i10 = 1.0f / (depth_at(d, Hs, Ws, H, W, y1, x0, a.lo, a.span) * ratio);
        const float i11 = 1.0f / (depth_at(d, Hs, Ws, H, W, y1, x1, a.lo, a.span) * ratio);
        const float inv = hy * (hx * i00 + lx * i01) + ly * (hx * i10 + lx * i11);
        o[0] = (inv - 0.01f) / 9.9f;
        int ch = 1;
        if (c.catxy) {
            // s-fold max-pool of depth * ratio = depth at the block's SMALLEST up-sampled disparity (depth is decreasing in it, and
            // rounding is monotone), times ratio
            float dmax = 0.f;
            for (int yy = rr * y; yy < rr * y + rr; ++yy)
                for (int xx = rr * x; xx < rr * x + rr; ++xx) dmax = fmaxf(dmax, depth_at(d, Hs, Ws, H, W, yy, xx, a.lo, a.span) * ratio);
            const float* k = a.inv_K[s] + b * 16;
            const float fx = (float)x, fy = (float)y;
            o[Ps] = dmax * (k[0] * fx + k[1] * fy + k[2]) / 30.0f;          // layers.py:190-200 (k_catxy's arithmetic)
            o[2 * Ps] = dmax * (k[4] * fx + k[5] * fy + k[6]) / 2.0f;
            o[3 * Ps] = (dmax * (k[8] * fx + k[9] * fy + k[10]) - 40.0f) / 40.0f;
            ch = 4;
        }
        const float* t2 = a.two_cha + (long)b * 2 * H * W;
        o[ch * Ps] = block_max(t2, H, W, y, x, rr);
        o[(ch + 1) * Ps] = block_max(t2 + (long)H * W, H, W, y, x, rr);
    }
}

int check_refine(const fd_refine_cfg* c, const char* who) {
    FD_REQUIRE(c, "%s: cfg is NULL", who);
    FD_REQUIRE(c->B > 0 && c->H > 0 && c->W > 0 && c->n_scales >= 1 && c->n_scales <= 4, "%s: bad sizes", who);
    FD_REQUIRE(c->min_depth > 0 && c->max_depth > c->min_depth, "%s: bad depth range", who);
    // an EMPTY window (y0 == y1 or x0 == x1) is legal: nothing is selected, the medians are NaN like torch.median of an empty selection
    FD_REQUIRE(0 <= c->crop_y0 && c->crop_y0 <= c->crop_y1 && c->crop_y1 <= c->H && 0 <= c->crop_x0 && c->crop_x0 <= c->crop_x1 && c->crop_x1 <= c->W,
               "%s: crop [%d,%d) x [%d,%d) outside %dx%d", who, c->crop_y0, c->crop_y1, c->crop_x0, c->crop_x1, c->H, c->W);
    for (int s = 0; s < c->n_scales; ++s) {
        const int r = c->Hs[s] > 0 ? c->H / c->Hs[s] : 0;
        FD_REQUIRE(r >= 1 && r <= 8 && (r & (r - 1)) == 0 && c->Hs[s] * r == c->H && c->Ws[s] * r == c->W,
                   "%s: scale %d (%dx%d) is not a power-of-two fraction (<= 8) of %dx%d", who, s, c->Hs[s], c->Ws[s], c->H, c->W);
        FD_REQUIRE(!c->pool_disp0 || r == (1 << s), "%s: pool_disp0 expects scale s at 1 / 2^s of the resolution", who);
    }
    FD_REQUIRE((long)c->B * c->H * c->W < (1L << 30), "%s: batch too large for 32-bit pixel indices", who);
    return 0;
}

inline long crop_cap(const fd_refine_cfg* c) { return (long)c->B * (c->crop_y1 - c->crop_y0) * (c->crop_x1 - c->crop_x0); }
inline long pooled_floats(const fd_refine_cfg* c) {
    long n = 0;
    if (c->pool_disp0)
        for (int s = 1; s < c->n_scales; ++s) n += (long)c->B * c->Hs[s] * c->Ws[s];
    return n;
}

struct MedArgs { const float* x; const float* gate; float scale; int B, H, W, y0, y1, x0, x1; int* idx; int* count; unsigned* keys; float* out; int cap; };
__global__ void __launch_bounds__(1024) k_masked_median(MedArgs a) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh[4];
    __shared__ int nan_seen;
    const int t = threadIdx.x;
    const int n = min(*a.count, a.cap);
    if (n == 0) { if (t == 0) { a.out[0] = __builtin_nanf(""); a.out[1] = 0.f; } return; }
    if (t == 0) nan_seen = 0;
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        const float v = a.x[a.idx[i]] * a.scale;
        if (v != v) nan_seen = 1;
        a.keys[i] = key_of(v);
    }
    __syncthreads();
    const unsigned key = select_rank(a.keys, n, (n - 1) / 2, hist, sh);
    if (t == 0) { a.out[0] = nan_seen ? __builtin_nanf("") : value_of(key); a.out[1] = (float)n; }
}

}  // namespace

extern "C" long fd_masked_median_ws_bytes(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return 0;
    return 16 + 8L * B * H * W;                 // counter + index list + keys
}

extern "C" int fd_masked_median(const float* x, const float* gate, float scale, int B, int H, int W, int y0, int y1, int x0, int x1,
                                float* out, void* ws, void* stream) {
    FD_REQUIRE(x && gate && out && ws && B > 0 && H > 0 && W > 0, "fd_masked_median: bad args");
    FD_REQUIRE(0 <= y0 && y0 < y1 && y1 <= H && 0 <= x0 && x0 < x1 && x1 <= W, "fd_masked_median: window outside the image");
    FD_REQUIRE((long)B * H * W < (1L << 30), "fd_masked_median: too many elements for 32-bit indices");
    hipStream_t st = (hipStream_t)stream;
    MedArgs a;
    a.x = x; a.gate = gate; a.scale = scale; a.B = B; a.H = H; a.W = W; a.y0 = y0; a.y1 = y1; a.x0 = x0; a.x1 = x1;
    a.count = (int*)ws; a.idx = (int*)((char*)ws + 16);
    a.cap = (int)((long)B * (y1 - y0) * (x1 - x0));
    a.keys = (unsigned*)(a.idx + (long)B * H * W);
    a.out = out;
    if (hipMemsetAsync(a.count, 0, 16, st) != hipSuccess) { fd_set_error("fd_masked_median: memset failed"); return -2; }
    const long n = a.cap;
    hipLaunchKernelGGL(k_refine_compact, dim3((unsigned)(n + 255) / 256 > 1024 ? 1024 : (unsigned)(n + 255) / 256), dim3(256), 0, st, gate, B, H, W,
                       y0, y1, x0, x1, a.idx, a.count);
    FD_LAUNCH_CHECK("fd_masked_median (compact)");
    hipLaunchKernelGGL(k_masked_median, dim3(1), dim3(1024), 0, st, a);
    FD_LAUNCH_CHECK("fd_masked_median");
    return 0;
}

extern "C" long fd_refine_inputs_ws_bytes(const fd_refine_cfg* c) {
    if (!c || c->B < 1 || c->n_scales < 1 || c->n_scales > 4) return 0;
    const long cap = crop_cap(c);
    return 64 + 4 * cap + 4 * cap * c->n_scales + 4 * pooled_floats(c);       // counter + ratios, indices, keys per scale, pooled disp_0
}

extern "C" int fd_refine_inputs(const fd_refine_cfg* c, const float* const* disp, const float* beam, const float* two_cha,
                                const float* const* inv_K, float* const* out, float* stats, void* ws, void* stream) {
    if (int rc = check_refine(c, "fd_refine_inputs")) return rc;
    FD_REQUIRE(disp && disp[0] && beam && two_cha && out && ws, "fd_refine_inputs: NULL argument");
    hipStream_t st = (hipStream_t)stream;
    RefArgs a;
    a.cfg = *c;
    a.lo = (float)(1.0 / c->max_depth); a.span = (float)(1.0 / c->min_depth - 1.0 / c->max_depth);
    a.beam = beam; a.two_cha = two_cha;
    const long cap = crop_cap(c);
    a.cap = (int)cap;
    char* w = (char*)ws;
    a.count = (int*)w; a.ratio = (float*)(w + 16);
    a.idx = (int*)(w + 64);
    a.keys = (unsigned*)(a.idx + cap);
    float* pooled = (float*)(a.keys + cap * c->n_scales);
    int nb = 0;
    for (int s = 0; s < 4; ++s) {
        const bool on = s < c->n_scales;
        a.disp[s] = nullptr; a.inv_K[s] = nullptr; a.out[s] = nullptr;
        a.first_block[s] = nb;
        if (!on) continue;
        FD_REQUIRE(out[s], "fd_refine_inputs: out[%d] is NULL", s);
        FD_REQUIRE(!c->catxy || (inv_K && inv_K[s]), "fd_refine_inputs: catxy needs inv_K[%d]", s);
        FD_REQUIRE(c->pool_disp0 || disp[s], "fd_refine_inputs: disp[%d] is NULL", s);
        a.inv_K[s] = c->catxy ? inv_K[s] : nullptr;
        a.out[s] = out[s];
        const long n = (long)c->B * c->Hs[s] * c->Ws[s];
        nb += (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
    }
    for (int s = c->n_scales; s < 5; ++s) a.first_block[s] = nb;
    if (hipMemsetAsync(w, 0, 64, st) != hipSuccess) { fd_set_error("fd_refine_inputs: memset failed"); return -2; }
    hipLaunchKernelGGL(k_refine_compact, dim3(cap / 256 + 1 > 1024 ? 1024 : (unsigned)(cap / 256 + 1)), dim3(256), 0, st, beam, c->B, c->H, c->W,
                       c->crop_y0, c->crop_y1, c->crop_x0, c->crop_x1, a.idx, a.count);
    FD_LAUNCH_CHECK("fd_refine_inputs (compact)");
    if (c->pool_disp0) {
        a.disp[0] = disp[0];
        if (c->n_scales > 1) {
            PoolArgs p;
            p.x = disp[0]; p.planes = c->B; p.H = c->H; p.W = c->W;
            int first = 0;
            float* q = pooled;
            for (int l = 0; l < 3; ++l) {
                p.first[l] = first;
                p.out[l] = nullptr; p.Hs[l] = p.Ws[l] = 0;
                if (l + 1 >= c->n_scales) continue;
                p.out[l] = q; p.Hs[l] = c->Hs[l + 1]; p.Ws[l] = c->Ws[l + 1];
                a.disp[l + 1] = q;
                const long n = (long)c->B * p.Hs[l] * p.Ws[l];
                q += n;
                first += (int)((n + 255) / 256);
            }
            p.first[3] = first;
            hipLaunchKernelGGL(k_refine_pool, dim3(first), dim3(256), 0, st, p);
            FD_LAUNCH_CHECK("fd_refine_inputs (pool)");
        }
    } else {
        for (int s = 0; s < c->n_scales; ++s) a.disp[s] = disp[s];
    }
    hipLaunchKernelGGL(k_refine_medians, dim3(c->n_scales), dim3(1024), 0, st, a);
    FD_LAUNCH_CHECK("fd_refine_inputs (medians)");
    hipLaunchKernelGGL(k_refine_outputs, dim3(nb), dim3(256), 0, st, a);
    FD_LAUNCH_CHECK("fd_refine_inputs (outputs)");
    if (stats && hipMemcpyAsync(stats, a.ratio, sizeof(float) * 4 * c->n_scales, hipMemcpyDeviceToDevice, st) != hipSuccess) {
        fd_set_error("fd_refine_inputs: stats copy failed");
        return -2;
    }
    return 0;
}
