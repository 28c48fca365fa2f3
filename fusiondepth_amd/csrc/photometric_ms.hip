// FD_HIPCC_FLAGS: -fno-slp-vectorize
// Multi-scale fused photometric-reprojection loss: ALL pyramid scales of generate_images_pred + compute_losses in one launch,
// forward value AND the unit-cotangent gradient in the same pass ("fd_photo_ms_*", include/fdhip.h).
//
// Reference path replaced: trainer.py:425-474 (bilinear upsample of disp, disp_to_depth, BackprojectDepth, Project3D,
// F.grid_sample border / bilinear / align_corners=False) and trainer.py:476-488, 509-567, 577-589 (SSIM + L1, identity losses +
// noise, per-pixel min, mean, masked SI-log LiDAR loss), for every scale of opt.scales.
//
// Why a second implementation next to photometric.hip (which stays for the flag variants, see fd_photo_ms_supported):
// the per-scale tile kernels are bound by the vector ALU, by LDS traffic and by re-warping halos (profiles/round1_pmc_loss.md:
// 1 030 + 2 360 VALU lane-instructions per pixel, 2.1-2.3x over-fetch).  Here
//   * a WAVE owns a strip of 64 image columns (60 of them outputs, 2 + 2 halo) and streams down its rows; lane = column, so
//     every HBM access of a row is one coalesced 256-byte segment and the 3x3 SSIM windows are separable sums:
//     horizontally two DPP wave shifts (v_add_f32_dpp wave_shr:1 / wave_shl:1, no LDS, no barrier), vertically a
//     three-row ring in registers;
//   * each pixel of a strip is warped ONCE per scale and frame (halo overhead 64/60 * (R+4)/R instead of the 1.33x re-warp of
//     the 16x64 tiles, and no second warp in a separate backward kernel);
//   * the two source frames of a strip run in the two waves of a 128-thread workgroup (half the live ring state per wave);
//     they exchange one loss value per pixel row through LDS for the 4-way argmin;
//   * the gradient w.r.t. the upsampled disparity for a unit upstream gradient (d mean / d disp_up) and the projection-matrix
//     gradients are produced in the same pass, two rows behind the warp front; the backward entry point only scales them by
//     the incoming gradients, adds the sparse LiDAR term and runs the adjoint of the bilinear upsampling;
//   * the SSIM is evaluated on window SUMS of values centred at 0.5 with every factor scaled by 81 (no divisions by 9, second
//     moments 4x smaller): max |error| vs a float64 evaluation 3e-6 against 6e-5 for the reference's own float32 arithmetic
//     (scripts/ssim_formulation_error.py).
// HBM-bound by design (169.2 B/pixel forward + 174.5 backward over the 4 scales); what bounds it in practice is
// the ~800 VALU lane-instructions per pixel and scale that remain.
#include "../../include/fdhip.h"
#include "fd_common.h"

namespace {

#ifndef FD_MS_WAVES
#define FD_MS_WAVES 2     // waves per SIMD the register allocation is held to
#endif
constexpr int OW = 60;             // output columns per strip (lanes 2..61)
constexpr float K1S = 81.0f * (float)(0.01 * 0.01), K2S = 81.0f * (float)(0.03 * 0.03);

__device__ __forceinline__ float shr1(float v) {   // lane l <- lane l-1 (0 at the wave edge)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float shl1(float v) {   // lane l <- lane l+1
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
#ifndef FD_MS_ABLATE
#define FD_MS_ABLATE 0    // timing experiments only (wrong results): 1 no barrier, 4 no DPP, 8 no LDS lag ring, 16 no LDS exchange
#endif
__device__ __forceinline__ float hsum3(float v) {
    if (FD_MS_ABLATE & 4) return (v + v * 0.99f) + v * 1.01f;
    return (v + shr1(v)) + shl1(v);
}

__device__ __forceinline__ int refl_clamp(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return fd_clampi(i, 0, n - 1);
}
// gfx950 issues fp32 add / mul / fma / mov on VGPR, inline-constant or literal operands at twice the rate of everything else
// (scripts/ubench/valu_rate2.hip: 2.6-2.9 vs 4.3-5.0 cycles per wave-instruction); ANY SGPR operand, and min / max / med3 /
// cmp / floor / cvt / shifts, are on the slow side.  Wave-uniform constants of the hot loop are therefore kept in VGPRs.
#ifndef FD_MS_EXACT_GRID
#define FD_MS_EXACT_GRID 0   // 1: normalise / unnormalise the sampling grid with the reference's operation sequence
#endif
#ifndef FD_MS_VCONST
#define FD_MS_VCONST 1
#endif
#ifndef FD_MS_PREFETCH
#define FD_MS_PREFETCH 1
#endif
__device__ __forceinline__ float vreg(float s) {
    float v = s;
    if (FD_MS_VCONST) asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ int vreg(int s) {
    int v = s;
    if (FD_MS_VCONST) asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

struct MsArgs {
    fd_photo_cfg cfg;        // B, H, W, NF (= 2), depth range, SI parameters, groups (Hs / Ws unused)
    int S, R;                // scales, rows per strip
    float lo, span;          // 1 / max_depth, 1 / min_depth - 1 / max_depth (host doubles -> float, as the reference's python floats)
    int Hs[4], Ws[4];
    int has_ident, want_grad;
    int nx, ny;              // strips per image along x / y
    unsigned beam_mask;      // bit s: scale s carries the LiDAR term
    const float* disp[4];
    const float* noise[4];   // [B,2,H,W] each, or NULL
    const float* inv_K; const float* P; const float* src[2]; const float* target; const float* ident; const float* beam;
    uint8_t* sel;            // [S,B,H,W]
    float* d1;               // [S,B,H,W]   d to_optimise.mean() / d disp_up (unit upstream gradient)
    float* part;             // [nblk][4]   sum(min), n_valid, sum(d), sum(d^2)
    float* gpart;            // [nblk][2][12]
};

// Lagged row data (written when a row is warped, read two rows later by the gradient stage) lives in LDS, 16 floats per
// lane and row: dX[3], dY[3], KX, KY, u, v, E0, E1, depth, x[3].  float4 groups, lane-contiguous (conflict-free b128 accesses).
struct Lag { float4 q[4]; };

// Everything of one image row that comes out of memory: issued one row ahead of its use (software prefetch; the gathers of row
// i+1 are in flight while the ~450 VALU instructions of row i run - a wave has no other independent work, and the register
// state allows only 2-3 waves per SIMD).
struct Pre {
    float nw[3], ne[3], sw[3], se[3], tg[3];
    float fx, fy, KX, KY, u, v, E0, E1, depth;
    float idv, nzv, bdv;        // identity candidate / noise / LiDAR value of the row ABOVE (consumed in the same step)
};

template <bool IDENT, bool GRAD>
__global__ void __launch_bounds__(128, FD_MS_WAVES) k_photo_ms(MsArgs a) {
    __shared__ float xl[2][2][2][64];        // [row parity][frame][loss | identity candidate][lane]
    __shared__ float xd[2][64];              // [row parity][lane]   frame-1 depth gradient
    __shared__ float4 lagr[2][3][4][64];     // [frame][row % 3][group][lane]
    const fd_photo_cfg& cfg = a.cfg;
    const int lane = threadIdx.x & 63;
    const int f = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int B = cfg.B, H = cfg.H, W = cfg.W;
    // 1-D grid, XCD-aware: workgroup L runs on XCD L % 8 (observed dispatch rule, used for speed only).  The S scales of one strip
    // read the same target / source / identity rows, so they get consecutive slots of ONE XCD - L = 8 S q + 8 s + r is scale s of
    // strip 8 q + r - and meet in that XCD's L2 instead of each missing to the fabric (FETCH_SIZE 2.4x -> see profiles/).
    const int L = blockIdx.x;
    const int q8 = L / (8 * a.S), r8 = L - q8 * (8 * a.S);
    const int s = r8 >> 3, strip = q8 * 8 + (r8 & 7);
    if (strip >= B * a.ny * a.nx) return;                    // padding of the last group of 8 strips (whole workgroup)
    const int b = strip / (a.ny * a.nx);
    const int by = (strip - b * a.ny * a.nx) / a.nx, bx = strip - (b * a.ny + by) * a.nx;
    const int P = H * W;
    const int Hs = a.Hs[s], Ws = a.Ws[s];
    const bool same = Hs == H && Ws == W;
    const int y0s = by * a.R;
    const int rows = min(a.R, H - y0s);
    const int n_iter = rows + 4;
    const bool has_beam = (a.beam_mask >> s) & 1u;

    // ---- per-lane invariants ----------------------------------------------------------------------------------------
    const int xs = bx * OW;
    const int cx = xs + lane - 2;
    const int gx = refl_clamp(cx, W);
    const bool colvalid = cx >= 0 && cx < W;
    const bool colown = lane >= 2 && lane < 62 && cx < W;
    const unsigned cxc = (unsigned)fd_clampi(cx, 0, W - 1);
    const float* iK = a.inv_K + b * 16;
    const float rayx0 = iK[0] * (float)gx + iK[2], rayx1 = iK[4] * (float)gx + iK[6], rayx2 = iK[8] * (float)gx + iK[10];
    const float iky0 = vreg(iK[1]), iky1 = vreg(iK[5]), iky2 = vreg(iK[9]);
    const float* Pf = a.P + ((long)b * 2 + f) * 12;
    float Pm[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) Pm[k] = vreg(Pf[k]);
    const float span_s = a.span;
    const float lo = vreg(a.lo), span = vreg(span_s);
    const float sWx = vreg((float)((double)W / (double)(W - 1))), sHy = vreg((float)((double)H / (double)(H - 1)));
    // clamp range of the sampling position: [1e-30, just below W-1].  A position that equals its clamped value is strictly inside
    // (aten clip_coordinates_set_grad treats the borders as outside); below W-1 the +1 taps always exist, and the sample differs
    // from the border pixel by < 1e-7 of the local contrast.
    const float xmm = vreg(__builtin_bit_cast(float, __builtin_bit_cast(int, (float)(W - 1)) - 1));
    const float ymm = vreg(__builtin_bit_cast(float, __builtin_bit_cast(int, (float)(H - 1)) - 1));
    const float epsv = vreg(cfg.eps);
    const float Wf = vreg((float)W);
    const int W4 = vreg(W * 4);
    const float inv_count = 1.0f / ((float)B * (float)H * (float)W);
    const float wS = (0.85f / 3.0f), wL = (0.15f / 3.0f);
    const float wSc = vreg(wS * inv_count), wLc = vreg(wL * inv_count);
    int x0d, x1d;
    float lxd;
    fd_bilinear_src(gx, (float)Ws / (float)W, Ws, x0d, x1d, lxd);
    const float hxd = 1.f - lxd;
    const float shd = (float)Hs / (float)H;
    // adjoint of ReflectionPad2d(1) along x: the window at column 0 also covers column -1 == column 1, so the 3-sum of column 1
    // counts column 0 twice (likewise W-1 -> W-2).  Only the strips that hold those columns take the weighted path.
    const bool xfold = xs == 0 || (xs <= W - 1 && W - 1 < xs + OW + 2);
    const float mshr = cx == 0 ? 2.f : 1.f;          // weight of this lane's value in its RIGHT neighbour's sum
    const float mshl = cx == W - 1 ? 2.f : 1.f;      // ... in its LEFT neighbour's sum

    const __amdgpu_buffer_rsrc_t r_src = fd_make_rsrc(a.src[f] + (long)b * 3 * P);
    const __amdgpu_buffer_rsrc_t r_tgt = fd_make_rsrc(a.target + (long)b * 3 * P);
    const __amdgpu_buffer_rsrc_t r_disp = fd_make_rsrc(a.disp[s] + (long)b * Hs * Ws);
    const bool has_noise = IDENT && a.noise[s];
    const __amdgpu_buffer_rsrc_t r_id = fd_make_rsrc(IDENT ? a.ident + ((long)b * 2 + f) * P : a.target);
    const __amdgpu_buffer_rsrc_t r_nz = fd_make_rsrc(has_noise ? a.noise[s] + ((long)b * 2 + f) * P : a.target);
    const __amdgpu_buffer_rsrc_t r_beam = fd_make_rsrc(has_beam ? a.beam + (long)b * P : a.target);
    const __amdgpu_buffer_rsrc_t r_sel = fd_make_rsrc(a.sel + ((long)s * B + b) * P);
    const __amdgpu_buffer_rsrc_t r_d1 = fd_make_rsrc(GRAD ? a.d1 + ((long)s * B + b) * P : a.part);
    const int cxc4 = (int)cxc * 4;

    // ---- streaming state (registers) ----------------------------------------------------------------------------------
    float y1r[3], y2r[3];                     // centred target of rows i-1, i-2
    float hp[15], hq[15];                     // horizontal 3-sums: previous row, (row before previous + previous row)
    float cp[9], cq[9];                       // same for the masked SSIM derivative coefficients (cq carries the y-fold weight)
    float l1_prev = 0.f, l1_cur = 0.f;        // L1 gradient weight of rows i-2 / i-1
    float lsum_prev = 0.f;                    // sum_c |y - x| of row i-1
#pragma unroll
    for (int c = 0; c < 3; ++c) y1r[c] = y2r[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 15; ++k) hp[k] = hq[k] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) cp[k] = cq[k] = 0.f;
    float gP[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gP[k] = 0.f;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;   // wave 0: sum(min); wave 1: n_valid, sum(d), sum(d^2)
    float pend = 0.f, pend_k = 0.f;                           // wave 0: its own depth gradient of the row in flight
    Pre pre[2];
    float dtap[2][5];                                         // disparity taps + row weight, issued two rows ahead

    // Source rows / weight of the bilinear upsampling (aten area_pixel_compute_source_index, align_corners=False).  For the
    // power-of-two pyramid ratios src = (2 dst + 1 - r) / 2r is exact in fp32, so the integer form on the scalar unit is bit-identical
    // to the float form (which costs ~12 half-rate VALU instructions per row for one wave-uniform value).
    const int rup = H / Hs;
    const bool pow2 = Hs * rup == H && (rup & (rup - 1)) == 0;
    const int sh2r = 31 - __builtin_clz(2 * rup);
    const float inv2r = 1.0f / (float)(2 * rup);
    auto src_rows = [&](const int gy, int& yy0, int& yy1, float& lyd) __attribute__((always_inline)) {
        if (pow2) {
            const int tnum = max(2 * gy + 1 - rup, 0);
            yy0 = min(tnum >> sh2r, Hs - 1);
            yy1 = yy0 + (yy0 < Hs - 1 ? 1 : 0);
            lyd = (float)(tnum & (2 * rup - 1)) * inv2r;
        } else {
            fd_bilinear_src(gy, shd, Hs, yy0, yy1, lyd);
            // wave-uniform, but computed with float VALU ops: without the readfirstlane hipcc wraps every load in a waterfall loop
            yy0 = __builtin_amdgcn_readfirstlane(yy0); yy1 = __builtin_amdgcn_readfirstlane(yy1);
        }
    };
    auto load_disp = [&](float (&da)[5], const int r) __attribute__((always_inline)) {
        const int gy = refl_clamp(y0s - 2 + r, H);
        if (same) {
            da[0] = ld(r_disp, gx * 4, gy * Ws * 4);
        } else {
            int yy0, yy1;
            src_rows(gy, yy0, yy1, da[4]);
            da[0] = ld(r_disp, x0d * 4, yy0 * Ws * 4); da[1] = ld(r_disp, x1d * 4, yy0 * Ws * 4);
            da[2] = ld(r_disp, x0d * 4, yy1 * Ws * 4); da[3] = ld(r_disp, x1d * 4, yy1 * Ws * 4);
        }
    };

    // ---- A: project row r, issue its gathers -------------------------------------------------------------------------------
    auto issue = [&](Pre& p, const float (&da)[5], const int r) __attribute__((always_inline)) {
        const int ry = y0s - 2 + r;
        const int gy = refl_clamp(ry, H);
#pragma unroll
        for (int c = 0; c < 3; ++c) p.tg[c] = ld(r_tgt, gx * 4, (c * P + gy * W) * 4);
        const int o1 = fd_clampi(ry - 1, 0, H - 1) * W * 4;
        p.idv = 0.f; p.nzv = 0.f; p.bdv = 0.f;
        if (IDENT) p.idv = ld(r_id, cxc4, o1);
        if (has_noise) p.nzv = ld(r_nz, cxc4, o1);
        if (has_beam && f == 1) p.bdv = ld(r_beam, cxc4, o1);
        float dup;
        if (same) {
            dup = da[0];
        } else {
            const float lyd = da[4];
            dup = (1.f - lyd) * (hxd * da[0] + lxd * da[1]) + lyd * (hxd * da[2] + lxd * da[3]);     // trainer.py:434-435
        }
        const float sdisp = fmaf(span, dup, lo);                                                  // layers.py:18-19
        const float rc0 = __builtin_amdgcn_rcpf(sdisp);
        const float depth = fmaf(fmaf(-rc0, sdisp, 1.0f), rc0, rc0);
        const float fy_ = (float)gy;
        const float ray0 = fmaf(iky0, fy_, rayx0), ray1 = fmaf(iky1, fy_, rayx1), ray2 = fmaf(iky2, fy_, rayx2);  // layers.py:158
        const float X0 = depth * ray0, X1 = depth * ray1, X2 = depth * ray2;
        const float c0 = fmaf(Pm[0], X0, fmaf(Pm[1], X1, fmaf(Pm[2], X2, Pm[3])));               // layers.py:219
        const float c1 = fmaf(Pm[4], X0, fmaf(Pm[5], X1, fmaf(Pm[6], X2, Pm[7])));
        const float c2 = fmaf(Pm[8], X0, fmaf(Pm[9], X1, fmaf(Pm[10], X2, Pm[11])));
        const float den = c2 + epsv;
        const float rc = __builtin_amdgcn_rcpf(den);
        float u = c0 * rc, v = c1 * rc;                                                           // layers.py:221
        u = fmaf(fmaf(-u, den, c0), rc, u);
        v = fmaf(fmaf(-v, den, c1), rc, v);
        // layers.py:224-226 + aten grid_sampler unnormalize (align_corners=False): ((2(u/(W-1) - .5) + 1) W - 1) / 2
#if FD_MS_EXACT_GRID
        const float gxn = (u / (float)(W - 1) - 0.5f) * 2.0f, gyn = (v / (float)(H - 1) - 0.5f) * 2.0f;
        const float ix = ((gxn + 1.0f) * (float)W - 1.0f) * 0.5f, iy = ((gyn + 1.0f) * (float)H - 1.0f) * 0.5f;
#else
        const float ix = fmaf(u, sWx, -0.5f), iy = fmaf(v, sHy, -0.5f);
#endif
        const float ixc = __builtin_amdgcn_fmed3f(ix, 1e-30f, xmm), iyc = __builtin_amdgcn_fmed3f(iy, 1e-30f, ymm);
        const float kx = ixc == ix ? sWx : 0.f;                  // clip_coordinates_set_grad: borders count as outside
        const float ky = iyc == iy ? sHy : 0.f;
        p.fx = __builtin_amdgcn_fractf(ixc); p.fy = __builtin_amdgcn_fractf(iyc);
        const float flx = ixc - p.fx, fly = iyc - p.fy;
        const int b00 = (int)(4.0f * fmaf(fly, Wf, flx));        // exact: integers below 2^24
        const int b01 = b00 + 4, b10 = b00 + W4, b11 = b10 + 4;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            p.nw[c] = ld(r_src, b00, c * P * 4); p.ne[c] = ld(r_src, b01, c * P * 4);
            p.sw[c] = ld(r_src, b10, c * P * 4); p.se[c] = ld(r_src, b11, c * P * 4);
        }
        const float Am0 = (c0 - Pm[3]) * sdisp, Am1 = (c1 - Pm[7]) * sdisp, Am2 = (c2 - Pm[11]) * sdisp;   // P[k,:3] . ray
        p.KX = kx * rc; p.KY = ky * rc; p.u = u; p.v = v;
        p.E0 = p.KX * fmaf(-u, Am2, Am0);
        p.E1 = p.KY * fmaf(-v, Am2, Am1);
        p.depth = depth;
    };

    // ---- B..G: everything that consumes row i -----------------------------------------------------------------------------------
    auto process = [&](const Pre& p, const int i) __attribute__((always_inline)) {
        const int ry = y0s - 2 + i, ry1 = ry - 1, ry2 = ry - 2;
        const int KC = i % 3, K1 = (i + 2) % 3, K2 = (i + 1) % 3;          // LDS slots of rows i, i-1, i-2
        Lag l2;                                                            // row i-2, consumed by the gradient stage at the end
        if (GRAD) {
#pragma unroll
            for (int k = 0; k < 4; ++k) l2.q[k] = (FD_MS_ABLATE & 8) ? make_float4(p.fx, p.fy, p.u, p.v) : lagr[f][K2][k][lane];
        }
        Lag lg;
        lg.q[1].z = p.KX; lg.q[1].w = p.KY; lg.q[2].x = p.u; lg.q[2].y = p.v; lg.q[2].z = p.E0; lg.q[2].w = p.E1; lg.q[3].x = p.depth;
        float hn[15], ycur[3];
        float lsum = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float dt = p.ne[c] - p.nw[c], db = p.se[c] - p.sw[c];
            const float top = fmaf(p.fx, dt, p.nw[c]), bot = fmaf(p.fx, db, p.sw[c]);
            const float dv = bot - top;
            const float pred = fmaf(p.fy, dv, top);
            const float dXv = fmaf(p.fy, db - dt, dt);
            const float xc = pred - 0.5f, yc = p.tg[c] - 0.5f;
            if (c == 0) { lg.q[0].x = dXv; lg.q[0].w = dv; lg.q[3].y = xc; }
            if (c == 1) { lg.q[0].y = dXv; lg.q[1].x = dv; lg.q[3].z = xc; }
            if (c == 2) { lg.q[0].z = dXv; lg.q[1].y = dv; lg.q[3].w = xc; }
            ycur[c] = yc;
            lsum += fabsf(yc - xc);
            hn[5 * c + 0] = hsum3(xc);
            hn[5 * c + 1] = hsum3(xc * xc);
            hn[5 * c + 2] = hsum3(xc * yc);
            hn[5 * c + 3] = hsum3(yc);
            hn[5 * c + 4] = hsum3(yc * yc);
        }
        if (GRAD && !(FD_MS_ABLATE & 8)) {
#pragma unroll
            for (int k = 0; k < 4; ++k) lagr[f][KC][k][lane] = lg.q[k];
        } else if (GRAD) {
            acc3 += lg.q[0].x + lg.q[0].y + lg.q[0].z + lg.q[0].w + lg.q[1].x + lg.q[1].y + lg.q[3].y + lg.q[3].z + lg.q[3].w;
        } else if (has_beam) {
            lagr[f][KC][3][lane] = lg.q[3];
        }
        // vertical 3-sums of row i-1: (h[i-2] + h[i-1]) + h[i]
        float S[15];
#pragma unroll
        for (int k = 0; k < 15; ++k) {
            S[k] = hq[k] + hn[k];
            hq[k] = hp[k] + hn[k];
            hp[k] = hn[k];
        }
        const float lsum1 = lsum_prev;
        lsum_prev = lsum;
        float yq3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { yq3[c] = y2r[c]; y2r[c] = y1r[c]; y1r[c] = ycur[c]; }
        // No early-outs for the first rows of a strip: every stage runs from row 0 on zero-initialised state (finite garbage that
        // is overwritten before it can reach an owned pixel); only the stores and the accumulators are masked.

        // ---- C: SSIM + L1 of row i-1, unmasked derivative coefficients -----------------------------------------------------
        const bool rowvalid1 = ry1 >= 0 && ry1 < H;
        float ssum = 0.f;
        float ta[3], tb[3], tc[3], wr[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float Cx = S[5 * c + 0], Cxx = S[5 * c + 1], Cxy = S[5 * c + 2], Cy = S[5 * c + 3], Cyy = S[5 * c + 4];
            const float Sx4 = Cx + 4.5f, Sy4 = Cy + 4.5f;                       // 9 mu_x, 9 mu_y   (layers.py:267-281 x 81)
            const float A1 = fmaf(Sx4 * Sy4, 2.0f, K1S);
            const float A2 = fmaf(Cx * Cy, -2.0f, fmaf(Cxy, 18.0f, K2S));
            const float B1 = fmaf(Sx4, Sx4, fmaf(Sy4, Sy4, K1S));
            const float B2 = fmaf(Cxx + Cyy, 9.0f, K2S) - fmaf(Cx, Cx, Cy * Cy);
            const float rd = __builtin_amdgcn_rcpf(B1 * B2);
            const float q = (A1 * A2) * rd;
            const float val = fmaf(q, -0.5f, 0.5f);
            const float valc = __builtin_amdgcn_fmed3f(val, 0.f, 1.f);
            ssum += valc;
            if (GRAD) {
                wr[c] = valc == val ? rd * wSc : 0.f;             // the clamp passes the gradient on [0, 1] (borders included)
                ta[c] = fmaf(q, fmaf(-Cx, B1, Sx4 * B2), fmaf(Cy, A1, -(Sy4 * A2)));   // d / d sum(x)           (x wr)
                tb[c] = 9.0f * (q * B1);                                                 // 2 x d / d sum(x^2)    (x wr)
                tc[c] = -9.0f * A1;                                                      // d / d sum(xy)         (x wr)
            }
        }
        const float Lown = fmaf(wS, ssum, wL * lsum1);           // trainer.py:476-488
        const float vown = IDENT ? fmaf(p.nzv, 0.00001f, p.idv) : 0.f;   // trainer.py:551-552
        // ---- D: exchange with the other frame's wave, 4-way argmin (trainer.py:549-567) ------------------------------------
        const int par = i & 1;
        if (!(FD_MS_ABLATE & 16)) {
            xl[par][f][0][lane] = Lown;
            if (IDENT) xl[par][f][1][lane] = vown;
        }
        if (!(FD_MS_ABLATE & 1)) __syncthreads();
        const float Loth = (FD_MS_ABLATE & 16) ? Lown * 1.01f : xl[par][1 - f][0][lane];
        // order of cat(identity -1, identity +1, reprojection -1, reprojection +1); the first minimum wins
        bool selown = f == 0 ? !(Loth < Lown) : (Lown < Loth);
        float best = fminf(Lown, Loth);
        float vmin = 0.f, voth = 0.f;
        if (IDENT) {
            voth = (FD_MS_ABLATE & 16) ? vown * 0.99f : xl[par][1 - f][1][lane];
            vmin = fminf(vown, voth);
            selown = selown && (Lown < vmin);
            best = fminf(best, vmin);
        }
        const bool rowown1 = i >= 3 && i < rows + 3;             // row index i-1 in [2, rows+2)
        const bool own1 = rowown1 && colown;
        if (f == 0) {
            if (own1) {
                acc0 += best;
                int bi;
                if (IDENT) bi = (vmin <= best) ? (voth < vown ? 1 : 0) : (selown ? 2 : 3);
                else bi = selown ? 0 : 1;
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bi, r_sel, cx, ry1 * W, 0);
            }
            if (GRAD && i >= 5) {   // D1 of row i-3: this wave's part is in `pend`, frame 1's arrived through xd
                const float other = xd[(i - 1) & 1][lane];
                if (colown) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (pend + other) * pend_k), r_d1, cx * 4, (ry - 3) * W * 4, 0);
            }
        } else if (has_beam) {                                      // trainer.py:577-589 / completor.py:718-723
            const float dep1 = lagr[f][K1][3][lane].x;
            const float d26 = dep1 * cfg.si_depth_scale;
            const float bd = p.bdv * cfg.si_beam_scale;
            bool m = own1 && bd > cfg.si_lo && d26 < 80.f && d26 > cfg.si_lo;
            if (cfg.si_mode == 0) m = m && fabsf(d26 - bd) < cfg.si_threshold;
            if (__any(m)) {
                if (m) {
                    if (cfg.si_mode == 1) { acc1 += 1.f; acc2 += fabsf(d26 - bd); }
                    else { const float d = logf(d26) - logf(bd); acc1 += 1.f; acc2 += d; acc3 += d * d; }
                }
            }
        }
        if (!GRAD) return;

        // ---- E: mask the coefficients by the selection, horizontal 3-sums -----------------------------------------------------
        const bool act = selown && rowvalid1 && colvalid;
        l1_prev = l1_cur;
        l1_cur = act ? wLc : 0.f;
        float cn[9];
        {
            float cf[9];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float w = act ? wr[c] : 0.f;
                cf[3 * c + 0] = w * ta[c]; cf[3 * c + 1] = w * tb[c]; cf[3 * c + 2] = w * tc[c];
            }
            if (xfold) {
#pragma unroll
                for (int j = 0; j < 9; ++j) cn[j] = (cf[j] + shr1(cf[j] * mshr)) + shl1(cf[j] * mshl);
            } else {
#pragma unroll
                for (int j = 0; j < 9; ++j) cn[j] = hsum3(cf[j]);
            }
        }
        // vertical 3-sums of the coefficient rows for image row i-2: c[i-3] + c[i-2] + c[i-1], row 0 counted twice for row 1
        // and row H-1 twice for row H-2 (adjoint of the reflection padding along y)
        const float wbot = vreg(ry2 == H - 2 ? 2.f : 1.f);       // weight of row i-1 in the sum for row i-2
        const float wtop = vreg(ry1 == 1 ? 2.f : 1.f);           // weight of row i-2 (== image row 0) in the NEXT row's sum
        float V[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            V[j] = fmaf(wbot, cn[j], cq[j]);
            cq[j] = fmaf(wtop, cp[j], cn[j]);
            cp[j] = cn[j];
        }

        // ---- F/G: gradient of row i-2 ---------------------------------------------------------------------------------
        const float dXq[3] = {l2.q[0].x, l2.q[0].y, l2.q[0].z}, dYq[3] = {l2.q[0].w, l2.q[1].x, l2.q[1].y};
        const float xq3[3] = {l2.q[3].y, l2.q[3].z, l2.q[3].w};
        float gix = 0.f, giy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float xq = xq3[c], yq = yq3[c];
            const float dq = xq - yq;
            const float sg = __builtin_amdgcn_fmed3f(dq * 1e30f, -l1_prev, l1_prev);   // l1w * sign(x - y)
            const float g = sg + fmaf(yq, V[3 * c + 2], fmaf(xq, V[3 * c + 1], V[3 * c + 0]));
            gix = fmaf(g, dXq[c], gix);
            giy = fmaf(g, dYq[c], giy);
        }
        if (!(colown && i >= 4 && i < n_iter)) { gix = 0.f; giy = 0.f; }   // row index i-2 must be in [2, rows+2)
        const float dd = fmaf(gix, l2.q[2].z, giy * l2.q[2].w);
        {
            const float ga = gix * l2.q[1].z, gb = giy * l2.q[1].w;
            const float gt = -fmaf(ga, l2.q[2].x, gb * l2.q[2].y);
            const float fy2 = (float)refl_clamp(ry2, H);
            const float dp = l2.q[3].x;
            const float Xa = dp * fmaf(iky0, fy2, rayx0), Xb = dp * fmaf(iky1, fy2, rayx1), Xc = dp * fmaf(iky2, fy2, rayx2);
            gP[0] = fmaf(ga, Xa, gP[0]); gP[1] = fmaf(ga, Xb, gP[1]); gP[2] = fmaf(ga, Xc, gP[2]); gP[3] += ga;
            gP[4] = fmaf(gb, Xa, gP[4]); gP[5] = fmaf(gb, Xb, gP[5]); gP[6] = fmaf(gb, Xc, gP[6]); gP[7] += gb;
            gP[8] = fmaf(gt, Xa, gP[8]); gP[9] = fmaf(gt, Xb, gP[9]); gP[10] = fmaf(gt, Xc, gP[10]); gP[11] += gt;
        }
        if (f == 0) {
            pend = dd;
            pend_k = -l2.q[3].x * l2.q[3].x * span_s;
        } else {
            xd[par][lane] = dd;
        }
    };

#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) lagr[f][k][j][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
#if FD_MS_PREFETCH
    load_disp(dtap[0], 0);
    load_disp(dtap[1], 1);
    issue(pre[0], dtap[0], 0);
#endif
    // two rows per trip (the prefetch buffers alternate with compile-time indices); an odd row count runs one masked extra row.
    // Rows past the strip are clamped by refl_clamp / fd_clampi: harmless loads, results unused.
    const int n_run = (n_iter + 1) & ~1;
#if FD_MS_PREFETCH
    // sched_barrier: hipcc otherwise sinks the prefetch loads down to their first use
    for (int i = 0; i < n_run; i += 2) {
        load_disp(dtap[0], i + 2);
        issue(pre[1], dtap[1], i + 1);
        __builtin_amdgcn_sched_barrier(0);
        process(pre[0], i);
        __builtin_amdgcn_sched_barrier(0);
        load_disp(dtap[1], i + 3);
        issue(pre[0], dtap[0], i + 2);
        __builtin_amdgcn_sched_barrier(0);
        process(pre[1], i + 1);
        __builtin_amdgcn_sched_barrier(0);
    }
#else
    for (int i = 0; i < n_run; ++i) {      // no software prefetch: fewer live registers, latency hidden by a third wave per SIMD
        load_disp(dtap[0], i);
        issue(pre[0], dtap[0], i);
        process(pre[0], i);
    }
#endif
    if (GRAD && n_run == n_iter) {
        __syncthreads();
        if (f == 0) {   // last owned row: index n_iter-3
            const float other = xd[(n_iter - 1) & 1][lane];
            if (colown) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (pend + other) * pend_k), r_d1, cx * 4, (y0s + rows - 1) * W * 4, 0);
        }
    }

    // ---- per-workgroup partial sums (fixed shuffle tree; combined in fixed order by k_photo_ms_fin) ----------------------------
    const long blk = (((long)s * B + b) * a.ny + by) * a.nx + bx;
    if (f == 0) {
        const float t = fd_wave_sum(acc0);
        if (lane == 0) a.part[blk * 4 + 0] = t;
    } else {
        const float t1 = fd_wave_sum(acc1), t2 = fd_wave_sum(acc2), t3 = fd_wave_sum(acc3);
        if (lane == 0) { a.part[blk * 4 + 1] = t1; a.part[blk * 4 + 2] = t2; a.part[blk * 4 + 3] = t3; }
    }
    if (GRAD) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const float t = fd_wave_sum(gP[k]);
            if (lane == 0) a.gpart[(blk * 2 + f) * 12 + k] = t;
        }
    }
}

// Blocks [0, S): the scalar loss terms of scale s (same `out` layout as fd_photo_fwd, FD_PHOTO_OUT_FLOATS floats per scale).
// Blocks [S, S + S*B): gP1[s][b][f][12] = sum over the strips of image b at scale s.
__global__ void __launch_bounds__(256) k_photo_ms_fin(const float* __restrict__ part, const float* __restrict__ gpart, int S, int B,
                                                      int groups, int blocks_per_image, float count, float si_var, unsigned beam_mask,
                                                      int si_mode, int want_grad, float* __restrict__ out, float* __restrict__ gP1) {
    __shared__ float red[4][4];
    __shared__ float gred[10][24];
    const int t = threadIdx.x;
    if ((int)blockIdx.x < S) {
        const int s = blockIdx.x;
        float* o = out + (long)s * FD_PHOTO_OUT_FLOATS;
        const int per_group = blocks_per_image * (B / groups);
        float tot = 0.f, si = 0.f;
        for (int g = 0; g < groups; ++g) {
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            const float* p = part + ((long)s * B * blocks_per_image + (long)g * per_group) * 4;
            for (int i = t; i < per_group; i += 256)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[k] += p[i * 4 + k];
            const float sum = fd_block_sum_n<4, 4>(acc, &red[0][0]);
            __shared__ float tt[4];
            if (t < 4) tt[t] = sum;
            __syncthreads();
            if (t == 0) {
                const float n = tt[1], m1 = tt[2] / n, m2 = tt[3] / n;
                const float var = m2 - si_var * (m1 * m1);
                const bool hb = (beam_mask >> s) & 1u;
                const float sl = hb ? (si_mode == 1 ? m1 * 0.001f : sqrtf(var) * 0.1f) : 0.f;
                o[8 + 4 * g] = n; o[9 + 4 * g] = m1; o[10 + 4 * g] = var; o[11 + 4 * g] = sl;
                tot += tt[0]; si += sl;
            }
            __syncthreads();
        }
        if (t == 0) {
            o[0] = tot / count;
            o[1] = o[8]; o[2] = o[9]; o[3] = o[10];
            o[4] = si / (float)groups;
            o[5] = 0.f; o[6] = 0.f; o[7] = 0.f;
        }
        return;
    }
    if (!want_grad) return;
    const int img = blockIdx.x - S;       // s * B + b
    const int k = t % 24, gi = t / 24;
    if (gi < 10) {
        float sacc = 0.f;
        for (int i = gi; i < blocks_per_image; i += 10) sacc += gpart[((long)img * blocks_per_image + i) * 24 + k];
        gred[gi][k] = sacc;
    }
    __syncthreads();
    if (t < 24) {
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) sacc += gred[j][t];
        gP1[(long)img * 24 + t] = sacc;
    }
}

// Backward: ONE launch for all scales.  d_up = g_photo[s] * D1 + LiDAR term, then the adjoint of the bilinear upsampling
// (trainer.py:434-435) for the scales below full resolution, then gP = sum_s g_photo[s] * gP1[s].
struct MsBwdArgs {
    fd_photo_cfg cfg;
    int S;
    float lo, span;
    int Hs[4], Ws[4];
    int first_block[5];                                        // block range of scale s: [first_block[s], first_block[s+1])
    int xchunks[4];                                            // column chunks per low-resolution row
    unsigned beam_mask;
    const float* disp[4];
    const float* beam; const float* stats;                     // stats [S][FD_PHOTO_OUT_FLOATS]
    const float* g_photo[4]; const float* g_si[4];             // one device float each, or NULL (= 0)
    const float* d1;                                           // [S,B,H,W]
    float* d_disp[4];
    const float* gP1; float* gP;                               // [S][B*24], [B*24]
};

struct SiTerm {      // per (scale, image): everything the LiDAR gradient of one pixel needs
    float k_si, k_l1, m1, lo, span;
};

// LiDAR part of d loss / d disp_up at full-resolution pixel (y, x) whose (scaled) LiDAR value bd passed the `bd > si_lo` gate
__device__ __forceinline__ float si_grad(const MsBwdArgs& a, const SiTerm& st, const float* __restrict__ disp_b, int Hs, int Ws, int y,
                                         int x, float bd) {
    const fd_photo_cfg& cfg = a.cfg;
    int y0, y1, x0, x1;
    float ly, lx;
    fd_bilinear_src(y, (float)Hs / (float)cfg.H, Hs, y0, y1, ly);
    fd_bilinear_src(x, (float)Ws / (float)cfg.W, Ws, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float dup = hy * (hx * disp_b[y0 * Ws + x0] + lx * disp_b[y0 * Ws + x1]) + ly * (hx * disp_b[y1 * Ws + x0] + lx * disp_b[y1 * Ws + x1]);
    const float sdisp = st.lo + st.span * dup;
    const float depth = 1.0f / sdisp;
    const float d26 = depth * cfg.si_depth_scale;
    float dd = 0.f;
    if (cfg.si_mode == 1) {
        if (d26 < 80.f && d26 > cfg.si_lo) dd = d26 > bd ? st.k_l1 : (d26 < bd ? -st.k_l1 : 0.f);
    } else if (d26 < 80.f && d26 > cfg.si_lo && fabsf(d26 - bd) < cfg.si_threshold) {
        const float dl = logf(d26) - logf(bd);
        dd = st.k_si * (dl - cfg.si_var * st.m1) * sdisp;
    }
    return -dd * depth * depth * st.span;
}
// d loss / d disp_up at pixel (y, x): photometric part (D1 holds it for a unit cotangent) + LiDAR part
__device__ __forceinline__ float up_grad(const MsBwdArgs& a, const SiTerm& st, const float* __restrict__ d1_b,
                                         const float* __restrict__ beam_b, const float* __restrict__ disp_b, int Hs, int Ws,
                                         float g_photo, int y, int x) {
    float v = g_photo * d1_b[y * a.cfg.W + x];
    if (beam_b) {
        const float bd = beam_b[y * a.cfg.W + x] * a.cfg.si_beam_scale;
        if (bd > a.cfg.si_lo) v += si_grad(a, st, disp_b, Hs, Ws, y, x, bd);
    }
    return v;
}

// Column sums of one low-resolution row iy for an even upsampling factor R_: the 2 R_ full-resolution rows are loaded first
// (independent loads in flight), then weighted.  Adjoint weights of the align_corners=False upsampling: output row
// R_ iy - R_/2 + k reads input row iy with weight 1 - |(k + 0.5) / R_ - 1|; rows clamped at the image border carry weight 1
// (both taps of aten's area_pixel_compute_source_index fall on the border pixel).
template <int R_>
__device__ __forceinline__ float column_sum(const MsBwdArgs& a, const SiTerm& st, const float* __restrict__ d1_b,
                                            const float* __restrict__ beam_b, const float* __restrict__ disp_b, int Hs, int Ws,
                                            float g_photo, int iy, int x) {
    const int H = a.cfg.H, W = a.cfg.W;
    const int oy0 = R_ * iy - R_ / 2;
    float v[2 * R_], bv[2 * R_];
#pragma unroll
    for (int k = 0; k < 2 * R_; ++k) {
        const int oy = oy0 + k;
        const bool ok = oy >= 0 && oy < H;
        v[k] = ok ? d1_b[oy * W + x] : 0.f;
        bv[k] = (ok && beam_b) ? beam_b[oy * W + x] : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 2 * R_; ++k) {
        const int oy = oy0 + k;
        const float d = ((float)k + 0.5f) * (1.0f / (float)R_) - 1.0f;
        float wy = 1.0f - fabsf(d);
        if ((iy == 0 && d < 0.f) || (iy == Hs - 1 && d > 0.f)) wy = 1.0f;
        float val = g_photo * v[k];
        const float bd = bv[k] * a.cfg.si_beam_scale;
        if (bd > a.cfg.si_lo) val += si_grad(a, st, disp_b, Hs, Ws, oy, x, bd);
        if (oy >= 0 && oy < H) acc += wy * val;
    }
    return acc;
}

// Block layout per scale s (r = H / Hs = W / Ws, an integer):
//   r == 1: grid-stride element-wise blocks (d_disp = d_up);
//   r  > 1: one block per (image, low-resolution row iy, chunk of TL = 256 / r - 1 low-resolution columns): the 2r
//           full-resolution rows that touch iy are reduced along y by 256 threads (one full-resolution column each, coalesced),
//           the column sums go through LDS and TL threads finish along x.
__global__ void __launch_bounds__(256) k_photo_ms_bwd(MsBwdArgs a) {
    __shared__ float colsum[256];
    const fd_photo_cfg& cfg = a.cfg;
    const int B = cfg.B, H = cfg.H, W = cfg.W;
    const int t = threadIdx.x;
    int blk = blockIdx.x;
    if (blk >= a.first_block[a.S]) {                 // gP blocks
        const int i = (blk - a.first_block[a.S]) * 256 + t;
        if (i < B * 24) {
            float acc = 0.f;
            for (int s = 0; s < a.S; ++s)
                if (a.g_photo[s]) acc += *a.g_photo[s] * a.gP1[(long)s * B * 24 + i];
            a.gP[i] = acc;
        }
        return;
    }
    int s = 0;
    while (s + 1 < a.S && blk >= a.first_block[s + 1]) ++s;
    blk -= a.first_block[s];
    const int Hs = a.Hs[s], Ws = a.Ws[s];
    const int r = H / Hs;
    const float g_photo = a.g_photo[s] ? *a.g_photo[s] : 0.f, g_si = a.g_si[s] ? *a.g_si[s] : 0.f;
    const bool has_beam = ((a.beam_mask >> s) & 1u) && g_si != 0.f;
    const long P = (long)H * W;
    int b, iy = 0, chunk = 0;
    if (r == 1) {
        const int per_img = a.xchunks[s];
        b = blk / per_img; chunk = blk - b * per_img;
    } else {
        const int per_img = Hs * a.xchunks[s];
        b = blk / per_img;
        const int rem = blk - b * per_img;
        iy = rem / a.xchunks[s]; chunk = rem - iy * a.xchunks[s];
    }
    SiTerm st;
    st.lo = a.lo; st.span = a.span;
    st.k_si = st.k_l1 = st.m1 = 0.f;
    if (has_beam) {
        const int grp = b / (B / cfg.groups);
        const float* so = a.stats + (long)s * FD_PHOTO_OUT_FLOATS;
        const float n_valid = so[8 + 4 * grp], var = so[10 + 4 * grp];
        st.m1 = so[9 + 4 * grp];
        st.k_si = g_si / (float)cfg.groups * 0.1f / (sqrtf(var) * n_valid);
        st.k_l1 = g_si / (float)cfg.groups * 0.001f / n_valid * cfg.si_depth_scale;
    }
    const float* d1_b = a.d1 + ((long)s * B + b) * P;
    const float* beam_b = has_beam ? a.beam + (long)b * P : nullptr;
    const float* disp_b = a.disp[s] + (long)b * Hs * Ws;
    float* out_b = a.d_disp[s] + (long)b * Hs * Ws;
    if (r == 1) {                                     // `chunk` strides over the image rows
        const int per_img = a.xchunks[s];
        for (int y = chunk; y < H; y += per_img)
            for (int x = t; x < W; x += 256) out_b[y * W + x] = up_grad(a, st, d1_b, beam_b, disp_b, Hs, Ws, g_photo, y, x);
        return;
    }
    const int TL = 256 / r - 1;
    const int ix0 = chunk * TL;
    const int cx0 = r * ix0 - r / 2;                  // first full-resolution column of this block
    const int x = cx0 + t;
    // Adjoint weights of the align_corners=False upsampling by an even integer factor r: output row r*iy - r/2 + k (k < 2r) reads
    // input row iy with weight 1 - |(k + 0.5) / r - 1|; rows / columns clamped at the image border carry weight 1 (both taps
    // of aten's area_pixel_compute_source_index fall on the border pixel).  Odd factors take the generic weights.
    const bool even = (r & 1) == 0;
    const float rinv = 1.0f / (float)r;
    float acc = 0.f;
    if (x >= 0 && x < W) {
        if (r == 2) acc = column_sum<2>(a, st, d1_b, beam_b, disp_b, Hs, Ws, g_photo, iy, x);
        else if (r == 4) acc = column_sum<4>(a, st, d1_b, beam_b, disp_b, Hs, Ws, g_photo, iy, x);
        else if (r == 8) acc = column_sum<8>(a, st, d1_b, beam_b, disp_b, Hs, Ws, g_photo, iy, x);
        else {
            const int oy_lo = max(0, r * iy - r / 2), oy_hi = min(H - 1, r * iy + r + r / 2 - 1);
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                float wy;
                if (even) {
                    const float d = ((float)(oy - (r * iy - r / 2)) + 0.5f) * rinv - 1.0f;
                    wy = 1.0f - fabsf(d);
                    if ((iy == 0 && d < 0.f) || (iy == Hs - 1 && d > 0.f)) wy = 1.0f;
                } else {
                    int y0, y1;
                    float ly;
                    fd_bilinear_src(oy, (float)Hs / (float)H, Hs, y0, y1, ly);
                    wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
                }
                if (wy != 0.f) acc += wy * up_grad(a, st, d1_b, beam_b, disp_b, Hs, Ws, g_photo, oy, x);
            }
        }
    }
    colsum[t] = acc;
    __syncthreads();
    const int ix = ix0 + t;
    if (t < TL && ix < Ws) {
        float o = 0.f;
        for (int k = 0; k < 2 * r; ++k) {
            const int col = r * t + k;                // relative to cx0
            const int ox = cx0 + col;
            if (ox < 0 || ox >= W) continue;
            float wx;
            if (even) {
                const float d = ((float)k + 0.5f) * rinv - 1.0f;
                wx = 1.0f - fabsf(d);
                if ((ix == 0 && d < 0.f) || (ix == Ws - 1 && d > 0.f)) wx = 1.0f;
            } else {
                int x0, x1;
                float lx;
                fd_bilinear_src(ox, (float)Ws / (float)W, Ws, x0, x1, lx);
                wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
            }
            o += wx * colsum[col];
        }
        out_b[iy * Ws + ix] = o;
    }
}

int check_ms(const fd_photo_ms_cfg* c, const char* who) {
    FD_REQUIRE(c, "%s: cfg is NULL", who);
    const fd_photo_cfg& b = c->base;
    FD_REQUIRE(b.B > 0 && b.H >= 4 && b.W >= 4, "%s: bad sizes B=%d H=%d W=%d", who, b.B, b.H, b.W);
    FD_REQUIRE(c->n_scales >= 1 && c->n_scales <= 4, "%s: n_scales must be 1..4 (got %d)", who, c->n_scales);
    for (int s = 0; s < c->n_scales; ++s)
        FD_REQUIRE(c->Hs[s] > 0 && c->Ws[s] > 0 && c->Hs[s] <= b.H && c->Ws[s] <= b.W, "%s: bad disparity size at scale %d", who, s);
    FD_REQUIRE(b.NF == 2 && b.use_ssim && !b.avg_reprojection,
               "%s: the multi-scale kernel covers NF == 2, SSIM, per-frame minimum (use fd_photo_fwd for the other variants)", who);
    FD_REQUIRE(b.groups >= 1 && b.groups <= 16 && b.B % b.groups == 0, "%s: batch %d not divisible into %d groups", who, b.B, b.groups);
    FD_REQUIRE(b.min_depth > 0 && b.max_depth > b.min_depth, "%s: bad depth range", who);
    FD_REQUIRE(b.si_mode == 0 || b.si_mode == 1, "%s: si_mode must be 0 or 1", who);
    FD_REQUIRE((long)b.B * b.H * b.W * 4 * c->n_scales < (1L << 31), "%s: batch too large for 32-bit offsets", who);
    return 0;
}

inline int ms_rows(const fd_photo_ms_cfg* c) {
    int R = c->rows_per_strip > 0 ? c->rows_per_strip : 40;     // measured at 192x640, batch 12 (scripts/ms_occupancy_sweep.sh): 16: 290 us, 32: 272, 40: 266, 48: 283, 64: 279, 96: 338
    return R > c->base.H ? c->base.H : R;
}
inline long ms_blocks_per_image(const fd_photo_ms_cfg* c) {
    return (long)fd_cdiv(c->base.W, OW) * fd_cdiv(c->base.H, ms_rows(c));
}

}  // namespace

extern "C" long fd_photo_ms_ws_floats(const fd_photo_ms_cfg* c) {
    if (!c || c->n_scales < 1 || c->base.B < 1) return 0;
    const long nblk = ms_blocks_per_image(c) * c->base.B * c->n_scales;
    return nblk * 4 + nblk * 24 + (long)c->n_scales * c->base.B * 24;
}

extern "C" int fd_photo_ms_fwd(const fd_photo_ms_cfg* c, const float* const* disp, const float* inv_K, const float* P,
                               const float* const* src, const float* target, const float* ident, const float* const* noise,
                               const float* beam, uint8_t* sel, float* d1, float* ws, float* out, void* stream) {
    if (int rc = check_ms(c, "fd_photo_ms_fwd")) return rc;
    FD_REQUIRE(disp && inv_K && P && src && src[0] && src[1] && target && sel && ws && out, "fd_photo_ms_fwd: NULL argument");
    FD_REQUIRE(!(c->beam_mask && !beam), "fd_photo_ms_fwd: beam_mask without beam");
    MsArgs a;
    a.cfg = c->base;
    a.S = c->n_scales; a.R = ms_rows(c);
    a.lo = (float)(1.0 / c->base.max_depth); a.span = (float)(1.0 / c->base.min_depth - 1.0 / c->base.max_depth);
    for (int s = 0; s < 4; ++s) {
        const bool on = s < c->n_scales;
        a.Hs[s] = on ? c->Hs[s] : 0; a.Ws[s] = on ? c->Ws[s] : 0;
        a.disp[s] = on ? disp[s] : nullptr;
        a.noise[s] = (on && noise && ident) ? noise[s] : nullptr;
        FD_REQUIRE(!on || disp[s], "fd_photo_ms_fwd: disp[%d] is NULL", s);
    }
    a.has_ident = ident ? 1 : 0;
    a.want_grad = d1 ? 1 : 0;
    a.beam_mask = beam ? c->beam_mask : 0u;
    a.inv_K = inv_K; a.P = P; a.src[0] = src[0]; a.src[1] = src[1]; a.target = target; a.ident = ident; a.beam = beam;
    a.sel = sel; a.d1 = d1;
    const long bpi = ms_blocks_per_image(c);
    const long nblk = bpi * c->base.B * c->n_scales;
    a.part = ws; a.gpart = ws + nblk * 4;
    float* gP1 = ws + nblk * 28;
    hipStream_t st = (hipStream_t)stream;
    a.nx = fd_cdiv(c->base.W, OW); a.ny = fd_cdiv(c->base.H, a.R);
    dim3 grid(fd_cdiv((long)c->base.B * a.ny * a.nx, 8) * 8 * c->n_scales);
    if (ident && d1) hipLaunchKernelGGL((k_photo_ms<true, true>), grid, dim3(128), 0, st, a);
    else if (ident) hipLaunchKernelGGL((k_photo_ms<true, false>), grid, dim3(128), 0, st, a);
    else if (d1) hipLaunchKernelGGL((k_photo_ms<false, true>), grid, dim3(128), 0, st, a);
    else hipLaunchKernelGGL((k_photo_ms<false, false>), grid, dim3(128), 0, st, a);
    FD_LAUNCH_CHECK("fd_photo_ms_fwd");
    const float count = (float)c->base.B * (float)c->base.H * (float)c->base.W;
    hipLaunchKernelGGL(k_photo_ms_fin, dim3(c->n_scales + (d1 ? c->n_scales * c->base.B : 0)), dim3(256), 0, st, a.part, a.gpart,
                       c->n_scales, c->base.B, c->base.groups, (int)bpi, count, c->base.si_var, a.beam_mask, c->base.si_mode,
                       a.want_grad, out, gP1);
    FD_LAUNCH_CHECK("fd_photo_ms_fin");
    return 0;
}

extern "C" int fd_photo_ms_bwd(const fd_photo_ms_cfg* c, const float* const* disp, const float* beam, const float* stats,
                               const float* const* g_photo, const float* const* g_si, float* d1, const float* ws,
                               float* const* d_disp, float* gP, void* stream) {
    if (int rc = check_ms(c, "fd_photo_ms_bwd")) return rc;
    FD_REQUIRE(disp && stats && g_photo && g_si && d1 && ws && d_disp && gP, "fd_photo_ms_bwd: NULL argument");
    MsBwdArgs a;
    a.cfg = c->base;
    a.S = c->n_scales;
    a.lo = (float)(1.0 / c->base.max_depth); a.span = (float)(1.0 / c->base.min_depth - 1.0 / c->base.max_depth);
    const int B = c->base.B, H = c->base.H, W = c->base.W;
    int nb = 0;
    for (int s = 0; s < 4; ++s) {
        const bool on = s < c->n_scales;
        a.Hs[s] = on ? c->Hs[s] : 0; a.Ws[s] = on ? c->Ws[s] : 0;
        a.disp[s] = on ? disp[s] : nullptr;
        a.g_photo[s] = on ? g_photo[s] : nullptr;
        a.g_si[s] = on ? g_si[s] : nullptr;
        a.d_disp[s] = on ? d_disp[s] : nullptr;
        a.first_block[s] = nb;
        a.xchunks[s] = 0;
        if (!on) continue;
        FD_REQUIRE(d_disp[s], "fd_photo_ms_bwd: d_disp[%d] is NULL", s);
        const int r = H / c->Hs[s];
        FD_REQUIRE(r >= 1 && r <= 16 && c->Hs[s] * r == H && c->Ws[s] * r == W,
                   "fd_photo_ms_bwd: scale %d (%dx%d) is not an integer fraction (<= 16) of %dx%d", s, c->Hs[s], c->Ws[s], H, W);
        if (r == 1) {
            a.xchunks[s] = H < 96 ? H : 96;          // row-strided blocks per image
            nb += B * a.xchunks[s];
        } else {
            a.xchunks[s] = fd_cdiv(c->Ws[s], 256 / r - 1);
            nb += B * c->Hs[s] * a.xchunks[s];
        }
    }
    for (int s = c->n_scales; s < 5; ++s) a.first_block[s] = nb;
    a.beam_mask = beam ? c->beam_mask : 0u;
    a.beam = beam; a.stats = stats; a.d1 = d1;
    const long nblk = ms_blocks_per_image(c) * B * c->n_scales;
    a.gP1 = ws + nblk * 28; a.gP = gP;
    nb += fd_cdiv(B * 24, 256);
    hipLaunchKernelGGL(k_photo_ms_bwd, dim3(nb), dim3(256), 0, (hipStream_t)stream, a);
    FD_LAUNCH_CHECK("fd_photo_ms_bwd");
    return 0;
}
