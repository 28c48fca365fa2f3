/* Oracle (test infrastructure only): C restatement of the sparse-LiDAR "2-channel" scatter,
 * reference gen2channel.py:60-117 (get_4beam_2channel).  Sequential, raster order, same update
 * rule as the reference so the result is bit-identical to it:
 *
 *   every non-zero LiDAR pixel (i,j) inside the ROI writes itself with confidence 1, then offers
 *   its depth to the cells at L1 distance dis = 1..expand that are NOT purely horizontal
 *   (|drow| = 1..dis, |dcol| = dis-|drow|) with confidence 1/(dis+1).  A cell takes an offer when
 *   it is empty or holds a lower confidence; offers of equal confidence are summed and counted;
 *   at the end each cell is divided by its count.
 *
 * Build: gcc -O2 -shared -fPIC -o liboracle_scatter.so scatter.c   (done by __graft_entry__.build)
 */
#include <stddef.h>

static void offer(float *depth, float *conf, float *cnt, int W, int r, int c, float v, float confidence)
{
    size_t p = (size_t)r * W + c;
    if (cnt[p] == 0.0f || conf[p] < confidence) {
        depth[p] = v;
        conf[p] = confidence;
        cnt[p] = 1.0f;
    } else if (conf[p] == confidence) {
        depth[p] += v;
        cnt[p] += 1.0f;
    }
}

/* beam: [H,W] fp32; depth_out, conf_out, scratch_cnt: [H,W] fp32 (all three are overwritten).
 * ROI rows [r0,r1), cols [c0,c1): 76..190 / 2..638 for the 192x640 trainer path
 * (gen2channel.py:64-65).  Returns 0, or -1 if the ROI + expand would leave the image. */
int fd_oracle_scatter_2channel(const float *beam, float *depth_out, float *conf_out, float *scratch_cnt,
                               int H, int W, int r0, int r1, int c0, int c1, int expand)
{
    if (r0 - expand < 0 || r1 - 1 + expand >= H || c0 - expand < 0 || c1 - 1 + expand >= W) return -1;
    for (size_t p = 0; p < (size_t)H * W; ++p) { depth_out[p] = 0.0f; conf_out[p] = 0.0f; scratch_cnt[p] = 0.0f; }
    for (int i = r0; i < r1; ++i) {
        for (int j = c0; j < c1; ++j) {
            float v = beam[(size_t)i * W + j];
            if (v == 0.0f) continue;
            size_t p = (size_t)i * W + j;
            depth_out[p] = v; conf_out[p] = 1.0f; scratch_cnt[p] = 1.0f;
            for (int dis = 1; dis <= expand; ++dis) {
                float confidence = (float)(1.0 / (dis + 1));
                for (int a = 1; a <= dis; ++a) {           /* a = |drow| >= 1: never purely horizontal */
                    int b = dis - a;                       /* b = |dcol| */
                    offer(depth_out, conf_out, scratch_cnt, W, i + a, j + b, v, confidence);
                    offer(depth_out, conf_out, scratch_cnt, W, i - a, j + b, v, confidence);
                    if (b != 0) {
                        offer(depth_out, conf_out, scratch_cnt, W, i + a, j - b, v, confidence);
                        offer(depth_out, conf_out, scratch_cnt, W, i - a, j - b, v, confidence);
                    }
                }
            }
        }
    }
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        float n = scratch_cnt[p] == 0.0f ? 1.0f : scratch_cnt[p];
        depth_out[p] = depth_out[p] / n;
    }
    return 0;
}
