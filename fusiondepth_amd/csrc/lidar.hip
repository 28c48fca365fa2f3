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
