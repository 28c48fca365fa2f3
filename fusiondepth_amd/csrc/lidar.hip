// Sparse-LiDAR "2-channel" scatter (reference gen2channel.py:60-117, get_4beam_2channel), as a GATHER:
// every output cell inspects its possible donors, so there are no write races and the float sums are
// formed in the same (raster) donor order as the reference's sequential loop => bit-identical output.
//
// Rule restated: a LiDAR return at (i,j) inside the ROI owns its cell (confidence 1) and offers its depth
// to the cells at L1 distance dis = 1..expand with |drow| >= 1 (never purely horizontal), confidence
// 1/(dis+1).  A cell keeps the highest confidence offered; equal-confidence offers are averaged.
#include "../../include/fdhip.h"
#include "fd_common.h"

namespace {

__global__ void __launch_bounds__(256) k_scatter2ch(const float* __restrict__ beam, float* __restrict__ out, int H, int W,
                                                    int r0, int r1, int c0, int c1, int expand) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const long P = (long)H * W;
    const float* bm = beam + b * P;
    float depth = 0.f, conf = 0.f;
    auto donor = [&](int r, int c) -> float {  // depth of a donating return at (r,c), 0 if none
        if (r < r0 || r >= r1 || c < c0 || c >= c1) return 0.f;
        return bm[(long)r * W + c];
    };
    const float own = donor(y, x);
    if (own != 0.f) {
        depth = own; conf = 1.0f;
    } else {
        for (int dis = 1; dis <= expand; ++dis) {
            float sum = 0.f, cnt = 0.f;
            for (int dr = -dis; dr <= dis; ++dr) {       // donor rows ascending = raster order
                if (dr == 0) continue;
                const int bcol = dis - (dr < 0 ? -dr : dr);
                for (int k = 0; k < (bcol ? 2 : 1); ++k) {
                    const int dc = bcol ? (k == 0 ? -bcol : bcol) : 0;
                    const float v = donor(y + dr, x + dc);
                    if (v != 0.f) {
                        sum = cnt == 0.f ? v : sum + v;
                        cnt += 1.f;
                    }
                }
            }
            if (cnt > 0.f) {
                depth = sum / cnt;
                conf = (float)(1.0 / (double)(dis + 1));
                break;
            }
        }
    }
    out[(long)b * 2 * P + (long)y * W + x] = depth;
    out[(long)b * 2 * P + P + (long)y * W + x] = conf;
}

}  // namespace

extern "C" int fd_scatter_2channel(const float* beam, float* out, int B, int H, int W, int r0, int r1, int c0, int c1,
                                   int expand, void* stream) {
    FD_REQUIRE(beam && out && B > 0 && H > 0 && W > 0, "fd_scatter_2channel: bad args");
    FD_REQUIRE(expand >= 1 && r0 - expand >= 0 && r1 - 1 + expand < H && c0 - expand >= 0 && c1 - 1 + expand < W,
               "fd_scatter_2channel: ROI [%d,%d)x[%d,%d) + expand %d leaves the %dx%d image", r0, r1, c0, c1, expand, H,
               W);
    dim3 grid(fd_cdiv(W, 64), fd_cdiv(H, 4), B);
    hipLaunchKernelGGL(k_scatter2ch, grid, dim3(256), 0, (hipStream_t)stream, beam, out, H, W, r0, r1, c0, c1, expand);
    FD_LAUNCH_CHECK("fd_scatter_2channel");
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// LiDAR rasterisation upstream of the scatter: Velodyne points -> z-buffered sparse depth image -> padded -> 2x2 max-pool
// -> "4beam" input.  Reference: kitti_utils.py:40-102 generate_depth_map, datasets/kitti_dataset.py:93-117 get_4beam,
// datasets/mono_dataset.py:193-198.  All arithmetic in float64 like the reference's numpy code; np.round = rint (ties to
// even).  The reference's duplicate handling is reproduced exactly (oracle/rasterize.py spells it out): a pixel holds the
// depth of the LAST point that hit it unless its `sub2ind` index (row * (W - 1) + col - 1) is shared by several points, in
// which case the pixel of the FIRST such point gets their minimum - and that index is also shared between column 0 of
// row r and column W - 1 of row r - 1.  Per-pixel min depth / first index / last index come from integer atomics (order
// independent), so the result is deterministic.
namespace {

__device__ __forceinline__ unsigned long long zkey(double z) {              // order-preserving map double -> u64
    const unsigned long long b = (unsigned long long)__double_as_longlong(z);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double zunkey(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

struct RasterWs {
    unsigned long long* zmin;   // [im_h * im_w]
    unsigned* first;            // [im_h * im_w]
    unsigned* last;             // [im_h * im_w]
    double* zs;                 // [n_points]
    double* depth;              // [im_h * im_w]
};

__global__ void k_raster_init(RasterWs w, long npix) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        w.zmin[i] = ~0ull; w.first[i] = 0xFFFFFFFFu; w.last[i] = 0u;
    }
}

__global__ void k_raster_points(const float* __restrict__ pts, int n, const double* __restrict__ P, int im_h, int im_w, int vel_depth,
                                RasterWs w) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double x = (double)pts[4 * i], y = (double)pts[4 * i + 1], z = (double)pts[4 * i + 2];
        if (!(x >= 0.0)) continue;                                         // kitti_utils.py:62
        // kitti_utils.py:65 np.dot(P_velo2im, velo.T): homogeneous coordinate 1.0
        const double c0 = fma(P[3], 1.0, fma(P[2], z, fma(P[1], y, P[0] * x)));
        const double c1 = fma(P[7], 1.0, fma(P[6], z, fma(P[5], y, P[4] * x)));
        const double c2 = fma(P[11], 1.0, fma(P[10], z, fma(P[9], y, P[8] * x)));
        const double u = rint(c0 / c2) - 1.0, v = rint(c1 / c2) - 1.0;     // :66, :73-74
        const double depth = vel_depth ? x : c2;                             // kitti_utils.py:68-69: forward distance instead of z
        w.zs[i] = depth;
        if (!(u >= 0.0 && v >= 0.0 && u < (double)im_w && v < (double)im_h)) continue;
        const long pix = (long)v * im_w + (long)u;
        atomicMin(&w.zmin[pix], zkey(depth));
        atomicMin(&w.first[pix], (unsigned)i);
        atomicMax(&w.last[pix], (unsigned)i);
    }
}

__global__ void k_raster_resolve(RasterWs w, int im_h, int im_w) {
    const long npix = (long)im_h * im_w;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long)gridDim.x * blockDim.x) {
        const int r = (int)(p / im_w), c = (int)(p - (long)r * im_w);
        double d = 0.0;
        if (w.first[p] != 0xFFFFFFFFu) {
            d = zunkey(w.zmin[p]);                                         // all points of this pixel share one sub2ind value
            // partner pixel with the same sub2ind value: (r, 0) <-> (r - 1, W - 1)
            long q = -1;
            if (im_w > 1 && c == 0 && r >= 1) q = (long)(r - 1) * im_w + (im_w - 1);
            else if (im_w > 1 && c == im_w - 1 && r + 1 < im_h) q = (long)(r + 1) * im_w;
            if (q >= 0 && w.first[q] != 0xFFFFFFFFu) {
                if (w.first[p] < w.first[q]) {                             // this pixel holds the group's first point: group minimum
                    const double dq = zunkey(w.zmin[q]);
                    d = dq < d ? dq : d;
                } else {
                    d = w.zs[w.last[p]];                                   // untouched by the duplicate pass: last point wins
                }
            }
            if (d < 0.0) d = 0.0;                                          // kitti_utils.py:86
        }
        w.depth[p] = d;
    }
}

// pad (top / left; optional 2-row crop) to the target shape: the float64 image generate_depth_map(shape=...) returns
__global__ void k_raster_pad(const double* __restrict__ depth, int im_h, int im_w, int pad_top, int pad_left, int crop_top, int tgt_h,
                             int tgt_w, double* __restrict__ out) {
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < (long)tgt_h * tgt_w; o += (long)gridDim.x * blockDim.x) {
        const int ty = (int)(o / tgt_w), tx = (int)(o - (long)ty * tgt_w);
        const int sy = ty + crop_top - pad_top, sx = tx - pad_left;
        out[o] = (sy >= 0 && sy < im_h && sx >= 0 && sx < im_w) ? depth[(long)sy * im_w + sx] : 0.0;
    }
}

// the same padding, then 2x2 max-pool with ceil_mode, float32, / 100
__global__ void k_raster_pool(const double* __restrict__ depth, int im_h, int im_w, int pad_top, int pad_left, int crop_top,
                              int tgt_h, int tgt_w, float* __restrict__ out, int out_h, int out_w) {
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < out_h * out_w; o += gridDim.x * blockDim.x) {
        const int oy = o / out_w, ox = o - oy * out_w;
        double m = -1.0 / 0.0;
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                const int ty = 2 * oy + dy, tx = 2 * ox + dx;
                if (ty >= tgt_h || tx >= tgt_w) continue;                  // ceil_mode: partial windows
                const int sy = ty + crop_top - pad_top, sx = tx - pad_left;
                const double v = (sy >= 0 && sy < im_h && sx >= 0 && sx < im_w) ? depth[(long)sy * im_w + sx] : 0.0;
                m = v > m ? v : m;
            }
        out[o] = (float)m / 100.0f;
    }
}

inline size_t align256(size_t n) { return (n + 255) / 256 * 256; }
}  // namespace

extern "C" long fd_velo_rasterize_ws_bytes(int n_points, int im_h, int im_w) {
    if (n_points < 0 || im_h <= 0 || im_w <= 0) return 0;
    const size_t npix = (size_t)im_h * im_w;
    return (long)(align256(npix * 8) + 2 * align256(npix * 4) + align256((size_t)(n_points > 0 ? n_points : 1) * 8) + align256(npix * 8));
}

extern "C" int fd_velo_rasterize(const float* points, int n_points, const double* P_velo2im, int im_h, int im_w, int vel_depth,
                                 int target_h, int target_w, float* beam_out, double* depth_out, void* ws, void* stream) {
    FD_REQUIRE((points || n_points == 0) && P_velo2im && (beam_out || depth_out) && ws && n_points >= 0 && im_h > 0 && im_w > 0,
               "fd_velo_rasterize: bad args");
    if (target_h <= 0 || target_w <= 0) { target_h = im_h; target_w = im_w; }   // shape=None: no padding
    FD_REQUIRE(target_w >= im_w, "fd_velo_rasterize: target width %d < image width %d (the reference pads, never crops, columns)",
               target_w, im_w);
    hipStream_t st = (hipStream_t)stream;
    const size_t npix = (size_t)im_h * im_w;
    char* b = (char*)ws;
    RasterWs w;
    w.zmin = (unsigned long long*)b; b += align256(npix * 8);
    w.first = (unsigned*)b; b += align256(npix * 4);
    w.last = (unsigned*)b; b += align256(npix * 4);
    w.zs = (double*)b; b += align256((size_t)(n_points > 0 ? n_points : 1) * 8);
    w.depth = (double*)b;
    hipLaunchKernelGGL(k_raster_init, dim3(fd_cdiv((long)npix, 256)), dim3(256), 0, st, w, (long)npix);
    FD_LAUNCH_CHECK("fd_velo_rasterize(init)");
    if (n_points > 0) {
        hipLaunchKernelGGL(k_raster_points, dim3(fd_cdiv(n_points, 256)), dim3(256), 0, st, points, n_points, P_velo2im, im_h, im_w,
                           vel_depth, w);
        FD_LAUNCH_CHECK("fd_velo_rasterize(points)");
    }
    hipLaunchKernelGGL(k_raster_resolve, dim3(fd_cdiv((long)npix, 256)), dim3(256), 0, st, w, im_h, im_w);
    FD_LAUNCH_CHECK("fd_velo_rasterize(resolve)");
    // kitti_utils.py:88-101: rows are padded on top by |target_h - im_h| (and 2 rows cropped when the target is shorter),
    // columns split the padding left / right
    const int ypad = target_h > im_h ? target_h - im_h : im_h - target_h;
    const int crop = target_h < im_h ? 2 : 0;
    const int padded_h = im_h + ypad - crop;
    FD_REQUIRE(padded_h >= 1, "fd_velo_rasterize: empty padded image");
    const int xpad1 = (target_w - im_w) / 2;
    if (depth_out) {
        hipLaunchKernelGGL(k_raster_pad, dim3(fd_cdiv((long)padded_h * target_w, 256)), dim3(256), 0, st, w.depth, im_h, im_w, ypad, xpad1,
                           crop, padded_h, target_w, depth_out);
        FD_LAUNCH_CHECK("fd_velo_rasterize(pad)");
    }
    if (beam_out) {
        const int out_h = (padded_h + 1) / 2, out_w = (target_w + 1) / 2;
        hipLaunchKernelGGL(k_raster_pool, dim3(fd_cdiv((long)out_h * out_w, 256)), dim3(256), 0, st, w.depth, im_h, im_w, ypad, xpad1, crop,
                           padded_h, target_w, beam_out, out_h, out_w);
        FD_LAUNCH_CHECK("fd_velo_rasterize(pool)");
    }
    return 0;
}
