// Geometry kernels of the hot path (reference layers.py:11-226) + library bookkeeping.
// All memory-bound, 1 thread per element / per batch item; written for wave64.
#include <stdarg.h>
#include <string.h>

#include "../../include/fdhip.h"
#include "fd_common.h"

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void fd_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" int fd_abi_version(void) { return FD_ABI_VERSION; }
extern "C" const char* fd_supported_arch(void) { return "gfx950"; }
extern "C" const char* fd_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// disp_to_depth  (layers.py:11-20)
__global__ void k_d2d_fwd(const float* __restrict__ disp, float* __restrict__ scaled, float* __restrict__ depth, long n,
                          float lo, float span) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = lo + span * disp[i];
        if (scaled) scaled[i] = s;
        if (depth) depth[i] = 1.0f / s;
    }
}
__global__ void k_d2d_bwd(const float* __restrict__ disp, const float* __restrict__ gs, const float* __restrict__ gd,
                          float* __restrict__ out, long n, float lo, float span) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float s = lo + span * disp[i];
        float g = 0.f;
        if (gs) g += gs[i];
        if (gd) g -= gd[i] / (s * s);
        out[i] = g * span;
    }
}
static inline int ew_grid(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
extern "C" int fd_disp_to_depth_fwd(const float* disp, float* scaled, float* depth, long n, double min_depth,
                                    double max_depth, void* stream) {
    FD_REQUIRE(disp && n >= 0, "fd_disp_to_depth_fwd: bad args");
    if (n == 0) return 0;
    float lo = (float)(1.0 / max_depth);
    float span = (float)(1.0 / min_depth - 1.0 / max_depth);
    hipLaunchKernelGGL(k_d2d_fwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, disp, scaled, depth, n, lo, span);
    FD_LAUNCH_CHECK("fd_disp_to_depth_fwd");
    return 0;
}
extern "C" int fd_disp_to_depth_bwd(const float* disp, const float* g_scaled, const float* g_depth, float* d_disp,
                                    long n, double min_depth, double max_depth, void* stream) {
    FD_REQUIRE(disp && d_disp && n >= 0, "fd_disp_to_depth_bwd: bad args");
    if (n == 0) return 0;
    float lo = (float)(1.0 / max_depth);
    float span = (float)(1.0 / min_depth - 1.0 / max_depth);
    hipLaunchKernelGGL(k_d2d_bwd, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, disp, g_scaled, g_depth, d_disp,
                       n, lo, span);
    FD_LAUNCH_CHECK("fd_disp_to_depth_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Pose vector -> 4x4 (layers.py:23-97).  One thread per batch item.
struct Rodrigues {
    float x, y, z, ca, sa, C, th, inv;  // axis, cos, sin, 1-cos, angle, 1/(angle+1e-7)
    float R[3][3];
};
__device__ __forceinline__ Rodrigues fd_rodrigues(float vx, float vy, float vz) {
    Rodrigues r;
    r.th = sqrtf(vx * vx + vy * vy + vz * vz);
    r.inv = 1.0f / (r.th + 1e-7f);
    r.x = vx * r.inv; r.y = vy * r.inv; r.z = vz * r.inv;
    r.ca = cosf(r.th); r.sa = sinf(r.th); r.C = 1.0f - r.ca;
    float xs = r.x * r.sa, ys = r.y * r.sa, zs = r.z * r.sa;
    float xC = r.x * r.C, yC = r.y * r.C, zC = r.z * r.C;
    float xyC = r.x * yC, yzC = r.y * zC, zxC = r.z * xC;
    r.R[0][0] = r.x * xC + r.ca; r.R[0][1] = xyC - zs;          r.R[0][2] = zxC + ys;
    r.R[1][0] = xyC + zs;        r.R[1][1] = r.y * yC + r.ca;   r.R[1][2] = yzC - xs;
    r.R[2][0] = zxC - ys;        r.R[2][1] = yzC + xs;          r.R[2][2] = r.z * zC + r.ca;
    return r;
}
__device__ __forceinline__ void pose_fwd_one(const float* __restrict__ aa, const float* __restrict__ tr, float* __restrict__ M, int invert) {
    Rodrigues r = fd_rodrigues(aa[0], aa[1], aa[2]);
    float t[3] = {tr[0], tr[1], tr[2]};
    if (!invert) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) M[i * 4 + j] = r.R[i][j];
            M[i * 4 + 3] = t[i];
        }
    } else {  // R^T * T(-t): rotation R^T, translation column sum_j R^T[i][j] * (-t_j)
        for (int i = 0; i < 3; ++i) {
            float acc = 0.f;
            for (int j = 0; j < 3; ++j) {
                M[i * 4 + j] = r.R[j][i];
                acc += r.R[j][i] * (-t[j]);
            }
            M[i * 4 + 3] = acc;
        }
    }
    M[12] = 0.f; M[13] = 0.f; M[14] = 0.f; M[15] = 1.f;
}
__global__ void k_pose_fwd(const float* __restrict__ aa, const float* __restrict__ tr, float* __restrict__ T, int B,
                           int invert) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    pose_fwd_one(aa + b * 3, tr + b * 3, T + b * 16, invert);
}
// G: the 4x4 gradient of the matrix, or NULL (the matrix was not used: zero gradients)
__device__ __forceinline__ void pose_bwd_one(const float* __restrict__ aa, const float* __restrict__ tr, const float* __restrict__ G,
                                             float* __restrict__ g_aa, float* __restrict__ g_tr, int invert) {
    float vx = aa[0], vy = aa[1], vz = aa[2];
    Rodrigues r = fd_rodrigues(vx, vy, vz);
    float t[3] = {tr[0], tr[1], tr[2]};
    float dR[3][3], dt[3];
    if (!invert) {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) dR[i][j] = G[i * 4 + j];
            dt[i] = G[i * 4 + 3];
        }
    } else {
        // M[i][j] = R[j][i];  M[i][3] = -sum_j R[j][i] t_j
        for (int j = 0; j < 3; ++j) {
            float acc = 0.f;
            for (int i = 0; i < 3; ++i) {
                dR[j][i] = G[i * 4 + j] - G[i * 4 + 3] * t[j];
                acc -= G[i * 4 + 3] * r.R[j][i];
            }
            dt[j] = acc;
        }
    }
    float x = r.x, y = r.y, z = r.z, C = r.C, sa = r.sa, ca = r.ca;
    float s01 = dR[0][1] + dR[1][0], s02 = dR[0][2] + dR[2][0], s12 = dR[1][2] + dR[2][1];
    float dx = dR[0][0] * 2.f * x * C + s01 * y * C + s02 * z * C + (dR[2][1] - dR[1][2]) * sa;
    float dy = dR[1][1] * 2.f * y * C + s01 * x * C + s12 * z * C + (dR[0][2] - dR[2][0]) * sa;
    float dz = dR[2][2] * 2.f * z * C + s02 * x * C + s12 * y * C + (dR[1][0] - dR[0][1]) * sa;
    float dC = dR[0][0] * x * x + dR[1][1] * y * y + dR[2][2] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
    float dca = dR[0][0] + dR[1][1] + dR[2][2] - dC;
    float dsa = -z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1];
    float dth = -sa * dca + ca * dsa;
    // axis = v * inv, inv = 1/(th + 1e-7)
    dth -= (dx * vx + dy * vy + dz * vz) * r.inv * r.inv;
    float k = r.th > 0.f ? dth / r.th : 0.f;  // d||v||/dv = v/||v|| (0 at the origin, as torch.norm)
    g_aa[0] = dx * r.inv + k * vx;
    g_aa[1] = dy * r.inv + k * vy;
    g_aa[2] = dz * r.inv + k * vz;
    g_tr[0] = dt[0]; g_tr[1] = dt[1]; g_tr[2] = dt[2];
}
__global__ void k_pose_bwd(const float* __restrict__ aa, const float* __restrict__ tr, const float* __restrict__ gT,
                           float* __restrict__ g_aa, float* __restrict__ g_tr, int B, int invert) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    pose_bwd_one(aa + b * 3, tr + b * 3, gT + b * 16, g_aa + b * 3, g_tr + b * 3, invert);
}

// The pose head of the stacked pose network (trainer.py:338-360 for all frame pairs and accumulated micro-batches in one go).
// pose [G * nf * Bq][ld]: the pose decoder's output (ld = 6 * frames-to-predict-for; the reference uses prediction 0 = columns 0..5:
// `axisangle[:, 0]`, `translation[:, 0]`), rows ordered (micro-batch g, frame pair k, sample s).  Per frame pair k: T_k [G * Bq][4][4]
// (inverted where bit k of invert_mask is set: trainer.py:352 `invert=(f_i < 0)`), and the pair's axisangle / translation
// [G * Bq][ld / 6][3] as the reference's outputs dictionary holds them.  One thread per (k, g, s).
struct PoseHeadArgs {
    const float* pose; float* g_pose;
    float* T[4]; const float* gT[4]; float* aa[4]; float* tr[4];
    int G, nf, Bq, ld;
    unsigned invert_mask;
};
__global__ void k_pose_head_fwd(PoseHeadArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, GB = a.G * a.Bq;
    if (i >= a.nf * GB) return;
    const int k = i / GB, j = i - k * GB, g = j / a.Bq, s = j - g * a.Bq;
    const float* row = a.pose + (long)((g * a.nf + k) * a.Bq + s) * a.ld;
    pose_fwd_one(row, row + 3, a.T[k] + j * 16, (a.invert_mask >> k) & 1u);
    const int nfp = a.ld / 6;
    if (a.aa[k])
        for (int f = 0; f < nfp; ++f)
            for (int c = 0; c < 3; ++c) {
                a.aa[k][(j * nfp + f) * 3 + c] = row[6 * f + c];
                a.tr[k][(j * nfp + f) * 3 + c] = row[6 * f + 3 + c];
            }
}
__global__ void k_pose_head_bwd(PoseHeadArgs a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, GB = a.G * a.Bq;
    if (i >= a.nf * GB) return;
    const int k = i / GB, j = i - k * GB, g = j / a.Bq, s = j - g * a.Bq;
    const long r = (long)((g * a.nf + k) * a.Bq + s) * a.ld;
    float* go = a.g_pose + r;
    if (a.gT[k]) pose_bwd_one(a.pose + r, a.pose + r + 3, a.gT[k] + j * 16, go, go + 3, (a.invert_mask >> k) & 1u);
    else for (int c = 0; c < 6; ++c) go[c] = 0.f;
    for (int c = 6; c < a.ld; ++c) go[c] = 0.f;              // predictions the reference never uses
}
extern "C" int fd_pose_matrix_fwd(const float* axisangle, const float* translation, float* T, int B, int invert,
                                  void* stream) {
    FD_REQUIRE(axisangle && translation && T && B > 0, "fd_pose_matrix_fwd: bad args");
    hipLaunchKernelGGL(k_pose_fwd, dim3(fd_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, axisangle, translation, T, B,
                       invert);
    FD_LAUNCH_CHECK("fd_pose_matrix_fwd");
    return 0;
}
extern "C" int fd_pose_matrix_bwd(const float* axisangle, const float* translation, const float* gT, float* g_axisangle,
                                  float* g_translation, int B, int invert, void* stream) {
    FD_REQUIRE(axisangle && translation && gT && g_axisangle && g_translation && B > 0, "fd_pose_matrix_bwd: bad args");
    hipLaunchKernelGGL(k_pose_bwd, dim3(fd_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, axisangle, translation, gT,
                       g_axisangle, g_translation, B, invert);
    FD_LAUNCH_CHECK("fd_pose_matrix_bwd");
    return 0;
}

extern "C" int fd_pose_head_fwd(const float* pose, float* const* T, float* const* axisangle, float* const* translation, int G, int nf,
                                int Bq, int ld, unsigned invert_mask, void* stream) {
    FD_REQUIRE(pose && T && G > 0 && nf > 0 && nf <= 4 && Bq > 0 && ld >= 6 && ld % 6 == 0, "fd_pose_head_fwd: bad args (1..4 frame pairs, ld = 6 * predictions)");
    FD_REQUIRE((axisangle == nullptr) == (translation == nullptr), "fd_pose_head_fwd: axisangle / translation come in pairs");
    PoseHeadArgs a = {};
    a.pose = pose; a.G = G; a.nf = nf; a.Bq = Bq; a.ld = ld; a.invert_mask = invert_mask;
    for (int k = 0; k < nf; ++k) {
        FD_REQUIRE(T[k], "fd_pose_head_fwd: T[%d] is NULL", k);
        a.T[k] = T[k];
        a.aa[k] = axisangle ? axisangle[k] : nullptr; a.tr[k] = translation ? translation[k] : nullptr;
        FD_REQUIRE((a.aa[k] == nullptr) == (a.tr[k] == nullptr), "fd_pose_head_fwd: axisangle / translation come in pairs");
    }
    hipLaunchKernelGGL(k_pose_head_fwd, dim3(fd_cdiv(nf * G * Bq, 64)), dim3(64), 0, (hipStream_t)stream, a);
    FD_LAUNCH_CHECK("fd_pose_head_fwd");
    return 0;
}
extern "C" int fd_pose_head_bwd(const float* pose, const float* const* gT, float* g_pose, int G, int nf, int Bq, int ld,
                                unsigned invert_mask, void* stream) {
    FD_REQUIRE(pose && gT && g_pose && G > 0 && nf > 0 && nf <= 4 && Bq > 0 && ld >= 6 && ld % 6 == 0, "fd_pose_head_bwd: bad args (1..4 frame pairs, ld = 6 * predictions)");
    PoseHeadArgs a = {};
    a.pose = pose; a.g_pose = g_pose; a.G = G; a.nf = nf; a.Bq = Bq; a.ld = ld; a.invert_mask = invert_mask;
    for (int k = 0; k < nf; ++k) a.gT[k] = gT[k];              // NULL: that matrix was not used
    hipLaunchKernelGGL(k_pose_head_bwd, dim3(fd_cdiv(nf * G * Bq, 64)), dim3(64), 0, (hipStream_t)stream, a);
    FD_LAUNCH_CHECK("fd_pose_head_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// P = (K @ T)[:3]   (layers.py:217)
__global__ void k_projmat_fwd(const float* __restrict__ K, const float* __restrict__ T, float* __restrict__ P,
                              long pstride, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 12) return;
    int b = i / 12, e = i % 12, r = e / 4, c = e % 4;
    const float* k = K + b * 16 + r * 4;
    const float* t = T + b * 16 + c;
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) acc += k[j] * t[j * 4];
    P[b * pstride + e] = acc;
}
__global__ void k_projmat_bwd(const float* __restrict__ K, const float* __restrict__ gP, long pstride,
                              float* __restrict__ gT, int B) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 16) return;
    int b = i / 16, e = i % 16, j = e / 4, c = e % 4;
    float acc = 0.f;
    for (int r = 0; r < 3; ++r) acc += K[b * 16 + r * 4 + j] * gP[b * pstride + r * 4 + c];
    gT[i] = acc;
}
extern "C" int fd_proj_matrix_fwd(const float* K, const float* T, float* P, long p_batch_stride, int B, void* stream) {
    FD_REQUIRE(K && T && P && B > 0 && p_batch_stride >= 12, "fd_proj_matrix_fwd: bad args");
    hipLaunchKernelGGL(k_projmat_fwd, dim3(fd_cdiv(B * 12, 64)), dim3(64), 0, (hipStream_t)stream, K, T, P,
                       p_batch_stride, B);
    FD_LAUNCH_CHECK("fd_proj_matrix_fwd");
    return 0;
}
extern "C" int fd_proj_matrix_bwd(const float* K, const float* gP, long p_batch_stride, float* gT, int B, void* stream) {
    FD_REQUIRE(K && gP && gT && B > 0 && p_batch_stride >= 12, "fd_proj_matrix_bwd: bad args");
    hipLaunchKernelGGL(k_projmat_bwd, dim3(fd_cdiv(B * 16, 64)), dim3(64), 0, (hipStream_t)stream, K, gP,
                       p_batch_stride, gT, B);
    FD_LAUNCH_CHECK("fd_proj_matrix_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// BackprojectDepth (layers.py:157-162)
__global__ void k_backproject_fwd(const float* __restrict__ depth, const float* __restrict__ invK,
                                  float* __restrict__ pts, int H, int W) {
    const int b = blockIdx.y;
    const long P = (long)H * W;
    const float* k = invK + b * 16;
    const float k00 = k[0], k01 = k[1], k02 = k[2], k10 = k[4], k11 = k[5], k12 = k[6], k20 = k[8], k21 = k[9],
                k22 = k[10];
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        float x = (float)(p % W), y = (float)(p / W);
        float d = depth[b * P + p];
        float* o = pts + (long)b * 4 * P + p;
        o[0] = d * (k00 * x + k01 * y + k02);
        o[P] = d * (k10 * x + k11 * y + k12);
        o[2 * P] = d * (k20 * x + k21 * y + k22);
        o[3 * P] = 1.0f;
    }
}
__global__ void k_backproject_bwd(const float* __restrict__ gpts, const float* __restrict__ invK,
                                  float* __restrict__ gdepth, int H, int W) {
    const int b = blockIdx.y;
    const long P = (long)H * W;
    const float* k = invK + b * 16;
    const float k00 = k[0], k01 = k[1], k02 = k[2], k10 = k[4], k11 = k[5], k12 = k[6], k20 = k[8], k21 = k[9],
                k22 = k[10];
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        float x = (float)(p % W), y = (float)(p / W);
        const float* g = gpts + (long)b * 4 * P + p;
        gdepth[b * P + p] = g[0] * (k00 * x + k01 * y + k02) + g[P] * (k10 * x + k11 * y + k12) +
                            g[2 * P] * (k20 * x + k21 * y + k22);
    }
}
extern "C" int fd_backproject_fwd(const float* depth, const float* inv_K, float* points, int B, int H, int W,
                                  void* stream) {
    FD_REQUIRE(depth && inv_K && points && B > 0 && H > 0 && W > 0, "fd_backproject_fwd: bad args");
    dim3 grid(ew_grid((long)H * W), B);
    hipLaunchKernelGGL(k_backproject_fwd, grid, dim3(256), 0, (hipStream_t)stream, depth, inv_K, points, H, W);
    FD_LAUNCH_CHECK("fd_backproject_fwd");
    return 0;
}
extern "C" int fd_backproject_bwd(const float* g_points, const float* inv_K, float* g_depth, int B, int H, int W,
                                  void* stream) {
    FD_REQUIRE(g_points && inv_K && g_depth && B > 0 && H > 0 && W > 0, "fd_backproject_bwd: bad args");
    dim3 grid(ew_grid((long)H * W), B);
    hipLaunchKernelGGL(k_backproject_bwd, grid, dim3(256), 0, (hipStream_t)stream, g_points, inv_K, g_depth, H, W);
    FD_LAUNCH_CHECK("fd_backproject_bwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Project3D (layers.py:215-226)
__device__ __forceinline__ void fd_load_P(const float* K, const float* T, float (&Pm)[12]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc += K[r * 4 + j] * T[j * 4 + c];
            Pm[r * 4 + c] = acc;
        }
}
__global__ void k_project_fwd(const float* __restrict__ pts, const float* __restrict__ K, const float* __restrict__ T,
                              float* __restrict__ grid, int H, int W, float eps) {
    const int b = blockIdx.y;
    const long P = (long)H * W;
    float Pm[12];
    fd_load_P(K + b * 16, T + b * 16, Pm);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        const float* x = pts + (long)b * 4 * P + p;
        float X0 = x[0], X1 = x[P], X2 = x[2 * P], X3 = x[3 * P];
        float c0 = Pm[0] * X0 + Pm[1] * X1 + Pm[2] * X2 + Pm[3] * X3;
        float c1 = Pm[4] * X0 + Pm[5] * X1 + Pm[6] * X2 + Pm[7] * X3;
        float c2 = Pm[8] * X0 + Pm[9] * X1 + Pm[10] * X2 + Pm[11] * X3;
        float den = c2 + eps;
        float u = c0 / den, v = c1 / den;
        float2 o;
        o.x = (u / wm1 - 0.5f) * 2.0f;
        o.y = (v / hm1 - 0.5f) * 2.0f;
        reinterpret_cast<float2*>(grid)[b * P + p] = o;
    }
}
// backward: g_points and per-block partial sums of gP (12 values)
__global__ void __launch_bounds__(256) k_project_bwd(const float* __restrict__ pts, const float* __restrict__ K,
                                                     const float* __restrict__ T, const float* __restrict__ ggrid,
                                                     float* __restrict__ gpts, float* __restrict__ part, int H, int W,
                                                     float eps) {
    __shared__ float red[4 * 12];
    const int b = blockIdx.y;
    const long P = (long)H * W;
    float Pm[12];
    fd_load_P(K + b * 16, T + b * 16, Pm);
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        const float* x = pts + (long)b * 4 * P + p;
        float X[4] = {x[0], x[P], x[2 * P], x[3 * P]};
        float c0 = Pm[0] * X[0] + Pm[1] * X[1] + Pm[2] * X[2] + Pm[3] * X[3];
        float c1 = Pm[4] * X[0] + Pm[5] * X[1] + Pm[6] * X[2] + Pm[7] * X[3];
        float c2 = Pm[8] * X[0] + Pm[9] * X[1] + Pm[10] * X[2] + Pm[11] * X[3];
        float den = c2 + eps;
        float u = c0 / den, v = c1 / den;
        float2 g = reinterpret_cast<const float2*>(ggrid)[b * P + p];
        float gu = g.x * 2.0f / wm1, gv = g.y * 2.0f / hm1;
        float dc[3] = {gu / den, gv / den, -(gu * u + gv * v) / den};
        float* go = gpts + (long)b * 4 * P + p;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            go[j * P] = Pm[j] * dc[0] + Pm[4 + j] * dc[1] + Pm[8 + j] * dc[2];
#pragma unroll
            for (int r = 0; r < 3; ++r) acc[r * 4 + j] += dc[r] * X[j];
        }
    }
    float s = fd_block_sum_n<12, 4>(acc, red);
    if (threadIdx.x < 12) part[((long)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = s;
}
__global__ void k_project_bwd_fin(const float* __restrict__ part, const float* __restrict__ K, float* __restrict__ gT,
                                  int nblk) {
    // one block of 64 threads per batch item: reduce gP over blocks (fixed order) then gT = K[:3]^T gP
    __shared__ float gP[12];
    const int b = blockIdx.x, t = threadIdx.x;
    if (t < 12) {
        float s = 0.f;
        for (int i = 0; i < nblk; ++i) s += part[((long)b * nblk + i) * 12 + t];
        gP[t] = s;
    }
    __syncthreads();
    if (t < 16) {
        int j = t / 4, c = t % 4;
        float a = 0.f;
        for (int r = 0; r < 3; ++r) a += K[b * 16 + r * 4 + j] * gP[r * 4 + c];
        gT[b * 16 + t] = a;
    }
}
static inline int proj_blocks(int H, int W) {
    int n = fd_cdiv((long)H * W, 256 * 4);
    return n < 1 ? 1 : (n > 256 ? 256 : n);
}
extern "C" long fd_project3d_bwd_ws_floats(int B, int H, int W) { return (long)B * proj_blocks(H, W) * 12; }
extern "C" int fd_project3d_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W,
                                float eps, void* stream) {
    FD_REQUIRE(points && K && T && grid && B > 0 && H > 1 && W > 1, "fd_project3d_fwd: bad args");
    dim3 g(ew_grid((long)H * W), B);
    hipLaunchKernelGGL(k_project_fwd, g, dim3(256), 0, (hipStream_t)stream, points, K, T, grid, H, W, eps);
    FD_LAUNCH_CHECK("fd_project3d_fwd");
    return 0;
}
extern "C" int fd_project3d_bwd(const float* points, const float* K, const float* T, const float* g_grid,
                                float* g_points, float* gT, float* ws, int B, int H, int W, float eps, void* stream) {
    FD_REQUIRE(points && K && T && g_grid && g_points && gT && ws && B > 0 && H > 1 && W > 1,
               "fd_project3d_bwd: bad args");
    int nb = proj_blocks(H, W);
    hipLaunchKernelGGL(k_project_bwd, dim3(nb, B), dim3(256), 0, (hipStream_t)stream, points, K, T, g_grid, g_points, ws,
                       H, W, eps);
    FD_LAUNCH_CHECK("fd_project3d_bwd");
    hipLaunchKernelGGL(k_project_bwd_fin, dim3(B), dim3(64), 0, (hipStream_t)stream, ws, K, gT, nb);
    FD_LAUNCH_CHECK("fd_project3d_bwd_fin");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Cat_xy (layers.py:187-201)
__global__ void k_catxy(const float* __restrict__ depth, const float* __restrict__ invK, float* __restrict__ out, int H,
                        int W) {
    const int b = blockIdx.y;
    const long P = (long)H * W;
    const float* k = invK + b * 16;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long)gridDim.x * blockDim.x) {
        float x = (float)(p % W), y = (float)(p / W);
        float d = depth[b * P + p];
        float* o = out + (long)b * 3 * P + p;
        o[0] = d * (k[0] * x + k[1] * y + k[2]) / 30.0f;
        o[P] = d * (k[4] * x + k[5] * y + k[6]) / 2.0f;
        o[2 * P] = (d * (k[8] * x + k[9] * y + k[10]) - 40.0f) / 40.0f;
    }
}
extern "C" int fd_cat_xy_fwd(const float* depth, const float* inv_K, float* out, int B, int H, int W, void* stream) {
    FD_REQUIRE(depth && inv_K && out && B > 0 && H > 0 && W > 0, "fd_cat_xy_fwd: bad args");
    hipLaunchKernelGGL(k_catxy, dim3(ew_grid((long)H * W), B), dim3(256), 0, (hipStream_t)stream, depth, inv_K, out, H,
                       W);
    FD_LAUNCH_CHECK("fd_cat_xy_fwd");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// cv2.resize(img, (Wout, Hout)) with INTER_LINEAR on a float32 image (evaluate_depth.py:349): OpenCV's coefficient rule
// (resize.cpp: scale in double, source coordinate rounded to float, floor, edge rule) and its two float32 passes, horizontal first.
// Not ATen's rule (k_bilinear_fwd computes the coordinate in float32 throughout): at x ~ 600 the two differ by ~6e-5 in the weight.
__device__ __forceinline__ void cv_linear_coeff(int d, double scale, int n_in, int& s0, int& s1, float& w0, float& w1) {
#pragma clang fp contract(off)     // no FMA contraction (HIP's __fmul_rn / __dmul_rn are plain operators): OpenCV's scalar arithmetic
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= n_in - 1) { s = n_in - 1; f = 0.f; }
    s0 = s; s1 = s + 1 < n_in ? s + 1 : n_in - 1;
    w0 = 1.0f - f; w1 = f;
}
__global__ void k_resize_linear_cv(const float* __restrict__ x, float* __restrict__ y, int Hin, int Win, int Hout, int Wout) {
    const int pl = blockIdx.y;
    const long Po = (long)Hout * Wout;
    const float* xi = x + (long)pl * Hin * Win;
    const double sy = (double)Hin / (double)Hout, sx = (double)Win / (double)Wout;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < Po; p += (long)gridDim.x * blockDim.x) {
#pragma clang fp contract(off)
        const int oy = (int)(p / Wout), ox = (int)(p % Wout);
        int y0, y1, x0, x1;
        float b0, b1, a0, a1;
        cv_linear_coeff(oy, sy, Hin, y0, y1, b0, b1);
        cv_linear_coeff(ox, sx, Win, x0, x1, a0, a1);
        // separate multiplies and adds (no contraction): the arithmetic of the numpy restatement in oracle/evaluate.py
        const float r0 = xi[y0 * Win + x0] * a0 + xi[y0 * Win + x1] * a1;
        const float r1 = xi[y1 * Win + x0] * a0 + xi[y1 * Win + x1] * a1;
        y[(long)pl * Po + p] = r0 * b0 + r1 * b1;
    }
}
extern "C" int fd_resize_linear_cv(const float* x, float* y, long planes, int Hin, int Win, int Hout, int Wout, void* stream) {
    FD_REQUIRE(x && y && planes > 0 && planes < 65536 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "fd_resize_linear_cv: bad args");
    hipLaunchKernelGGL(k_resize_linear_cv, dim3(ew_grid((long)Hout * Wout), (unsigned)planes), dim3(256), 0, (hipStream_t)stream, x, y, Hin, Win,
                       Hout, Wout);
    FD_LAUNCH_CHECK("fd_resize_linear_cv");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// bilinear resize, align_corners=False (trainer.py:434-435)
__global__ void k_bilinear_fwd(const float* __restrict__ x, float* __restrict__ y, int Hin, int Win, int Hout, int Wout,
                               float sh, float sw) {
    const int bc = blockIdx.y;
    const long Po = (long)Hout * Wout;
    const float* xi = x + (long)bc * Hin * Win;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < Po; p += (long)gridDim.x * blockDim.x) {
        int oy = (int)(p / Wout), ox = (int)(p % Wout);
        int y0, y1, x0, x1;
        float ly, lx;
        fd_bilinear_src(oy, sh, Hin, y0, y1, ly);
        fd_bilinear_src(ox, sw, Win, x0, x1, lx);
        float hy = 1.f - ly, hx = 1.f - lx;
        y[(long)bc * Po + p] = hy * (hx * xi[y0 * Win + x0] + lx * xi[y0 * Win + x1]) +
                               ly * (hx * xi[y1 * Win + x0] + lx * xi[y1 * Win + x1]);
    }
}
// Adjoint in gather form: each input pixel collects from the output pixels whose 2x2 footprint holds it.  The bilinear
// weight factorises, w(oy, ox) = wy(oy) * wx(ox): the column weights of the candidate window are computed once per
// thread (<= MAXW = 24 candidates for upsampling ratios <= 8, the four pyramid scales), then every candidate row costs one weight + a short dot product.
__global__ void k_bilinear_bwd(const float* __restrict__ gy, float* __restrict__ gx, int Hin, int Win, int Hout,
                               int Wout, float sh, float sw) {
    constexpr int MAXW = 24;
    const int bc = blockIdx.y;
    const int Pi = Hin * Win;
    const float* g = gy + (long)bc * Hout * Wout;
    const int ry = (Hout + Hin - 1) / Hin, rx = (Wout + Win - 1) / Win;  // upsampling ratio (ceil)
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < Pi; p += gridDim.x * blockDim.x) {
        const int iy = p / Win, ix = p - iy * Win;
        // output rows whose source interval [y0, y1] contains iy satisfy (iy-0.5)*r - 0.5 <= oy < (iy+1.5)*r - 0.5
        // (before edge clamping, which only adds rows at the borders handled by the max/min below); one extra on each side
        int oy_lo = (2 * iy - 1) * ry / 2 - 2, oy_hi = (2 * iy + 3) * ry / 2 + 1;
        int ox_lo = (2 * ix - 1) * rx / 2 - 2, ox_hi = (2 * ix + 3) * rx / 2 + 1;
        if (iy == 0) oy_lo = 0;
        if (ix == 0) ox_lo = 0;
        if (iy == Hin - 1) oy_hi = Hout - 1;
        if (ix == Win - 1) ox_hi = Wout - 1;
        oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
        oy_hi = oy_hi > Hout - 1 ? Hout - 1 : oy_hi; ox_hi = ox_hi > Wout - 1 ? Wout - 1 : ox_hi;
        float acc = 0.f;
        const int nx = ox_hi - ox_lo + 1;
        if (nx <= MAXW) {
            float wxs[MAXW];
#pragma unroll
            for (int k = 0; k < MAXW; ++k) {
                int x0, x1; float lx;
                const int ox = ox_lo + k < Wout ? ox_lo + k : Wout - 1;
                fd_bilinear_src(ox, sw, Win, x0, x1, lx);
                const float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                wxs[k] = k < nx ? wx : 0.f;
            }
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int y0, y1; float ly;
                fd_bilinear_src(oy, sh, Hin, y0, y1, ly);
                const float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
                if (wy == 0.f) continue;
                const float* row = g + (long)oy * Wout + ox_lo;
                // same accumulation order as the scalar form below: ox ascending, zero weights skipped
#pragma unroll
                for (int k = 0; k < MAXW; ++k)
                    if (wxs[k] != 0.f) acc += wy * wxs[k] * row[k < nx ? k : 0];
            }
        } else {
            for (int oy = oy_lo; oy <= oy_hi; ++oy) {
                int y0, y1; float ly;
                fd_bilinear_src(oy, sh, Hin, y0, y1, ly);
                float wy = (y0 == iy ? 1.f - ly : 0.f) + (y1 == iy ? ly : 0.f);
                if (wy == 0.f) continue;
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    int x0, x1; float lx;
                    fd_bilinear_src(ox, sw, Win, x0, x1, lx);
                    float wx = (x0 == ix ? 1.f - lx : 0.f) + (x1 == ix ? lx : 0.f);
                    if (wx != 0.f) acc += wy * wx * g[(long)oy * Wout + ox];
                }
            }
        }
        gx[(long)bc * Pi + p] = acc;
    }
}
extern "C" int fd_bilinear_up_fwd(const float* x, float* y, int BC, int Hin, int Win, int Hout, int Wout, void* stream) {
    FD_REQUIRE(x && y && BC > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "fd_bilinear_up_fwd: bad args");
    hipLaunchKernelGGL(k_bilinear_fwd, dim3(ew_grid((long)Hout * Wout), BC), dim3(256), 0, (hipStream_t)stream, x, y, Hin,
                       Win, Hout, Wout, (float)Hin / (float)Hout, (float)Win / (float)Wout);
    FD_LAUNCH_CHECK("fd_bilinear_up_fwd");
    return 0;
}
extern "C" int fd_bilinear_up_bwd(const float* gy, float* gx, int BC, int Hin, int Win, int Hout, int Wout,
                                  void* stream) {
    FD_REQUIRE(gy && gx && BC > 0 && Hin > 0 && Win > 0 && Hout >= Hin && Wout >= Win,
               "fd_bilinear_up_bwd: bad args (only upsampling is supported)");
    hipLaunchKernelGGL(k_bilinear_bwd, dim3(ew_grid((long)Hin * Win), BC), dim3(256), 0, (hipStream_t)stream, gy, gx, Hin,
                       Win, Hout, Wout, (float)Hin / (float)Hout, (float)Win / (float)Wout);
    FD_LAUNCH_CHECK("fd_bilinear_up_bwd");
    return 0;
}
