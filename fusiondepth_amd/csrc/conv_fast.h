// Internal interface between conv.hip (C-ABI entry points, shape logic) and conv_fast.hip (fast-path kernels).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fdhip.h"

struct FastGemmArgs {
    const float* A;      // [M][T][C] re-laid-out weights
    const float* X;      // gathered tensor [Nb][C][Hi][Wi]
    float* Y;            // output (affine pixel map below)
    const float* bias;   // [M] or null
    float* slabs;        // split-K workspace (splits * out_total floats) or null
    long slab_stride, out_total;
    int M, K;            // K = T * C
    int Nb, C, Hi, Wi;
    int NY, NX;          // GEMM-N domain per image
    int T, TB;           // taps, taps per row
    int sy, oy, da, sx, ox, db;
    int pad_mode;
    long out_ns, out_cs;
    int out_w, osy, ooy, osx, oox;
    int act;
    int xcd_swizzle;
    const float* add;    // optional, laid out like Y: Y = act(conv + bias) + add  (a second gradient arriving at the same tensor:
                         // the residual branch of a ResNet block joins the data gradient in the epilogue)
};

// the fields in which the problems of one grouped launch differ (fast_gemm_group_launch)
struct FastGemmGroup {
    const float* A[4];
    int NY[4], NX[4], oy[4], ox[4], ooy[4], oox[4], T[4], TB[4], K[4];
    // output map per problem (used when own_out != 0: the problems write to different places of the buffer Y, e.g. the four sides
    // of a padded grid's ring): Y + y_off[j] floats, row width out_w[j], channel stride out_cs[j], image stride out_ns[j]
    long y_off[4], out_ns[4], out_cs[4];
    int out_w[4];
    int own_out;
    int first_bx[5];
    int n;
};

struct FastWgradArgs {
    const float* dY; const float* X; float* slabs;   // slabs [splits][M][T][C]
    int M, C, T, TB;
    int Nb, Hi, Wi, NY, NX;
    int sy, oy, da, sx, ox, db;
    int pad_mode;
    long dy_ns, dy_cs;
    long pix_per_split;
};

long fast_splitk_slab_floats(const FastGemmArgs& a, int* splits_out);
int fast_gemm_launch(const FastGemmArgs& a, hipStream_t st);
int fast_gemm_group_launch(const FastGemmArgs& a, const FastGemmGroup& q, hipStream_t st);
int fast_weight_relayout(const float* W, float* A2, int Co, int Ci, int KH, int KW, int TA, int TB, int kh0, int dkh, int kw0,
                         int dkw, int mode, hipStream_t st);
int fast_wgrad_splits(int M, int C, int T, long Np);
int fast_wgrad_launch(const FastWgradArgs& a, float* gw, int splits, int accumulate, hipStream_t st);

// Y[i] = act(sum_z slabs[z][i] + bias[channel(i)]) - the deterministic split-K epilogue (also used by conv_wino.hip)
int fast_splitk_finish_launch(const float* slabs, float* Y, const float* bias, long total, long slab_stride, int splits, long out_cs,
                              int M, int act, hipStream_t st, const float* add = nullptr);

// conv_wino.hip: 3x3 stride-1 convolutions through the 1-D Winograd F(2,3) transform
bool wino_fwd_ok(const fd_conv_desc* d);
long wino_wt_floats(const fd_conv_desc* d);
bool wino_fwd_2d(const fd_conv_desc* d);
bool wino_fwd_limb(const fd_conv_desc* d);      // k_conv_wino2d_limb: the weight layout is the limb image of U2 (re-layout modes 11 / 12)
long wino_ws_floats(const fd_conv_desc* d);
int wino_weight_launch(const fd_conv_desc* d, const float* w, float* U, int flip, hipStream_t st);
// the BatchNorm that follows a slab-route convolution, fused with the slab reduction (norm.hip: k_bn_train_small_slabs)
struct BnAfterConv {
    const float* weight; const float* bias; const float* residual; float* out;
    float* running_mean; float* running_var; float* save_mean; float* save_invstd;
    int groups; float eps, momentum; int relu;
};
bool bn_small_slabs_ok(int N, int C, int H, int W, int groups);
int bn_small_slabs_launch(const float* slabs, long slab_stride, int ksplit, float* y, const BnAfterConv& bn, int N, int C, int H, int W,
                          hipStream_t st);
bool wino_fwd_slab_route(const fd_conv_desc* d);
int wino_conv_launch(const fd_conv_desc* d, const float* x, const float* U, const float* bias, float* y, float* ws, hipStream_t st,
                     const float* add = nullptr, float* stat_part = nullptr, const BnAfterConv* bn = nullptr);
int wino_stat_slots(const fd_conv_desc* d);
// conv_n16.hip: 3x3 stride-1 convolutions with 16 / 32 channels on either side (the decoder's full-resolution blocks)
bool n16_shape_ok(const fd_conv_desc* d, int M, int C);
int n16_launch(const fd_conv_desc* d, int M, int C, const float* x, const float* w, const float* bias, float* y, int flip,
               int pad_mode, int act, hipStream_t st);
// conv_c1.hip: 3x3 stride-1 convolutions with one output channel (dispconv) as stencils
bool c1_shape_ok(const fd_conv_desc* d);
int c1_fwd_launch(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st);
int c1_dgrad_launch(const fd_conv_desc* d, const float* gy, const float* w, float* gx, hipStream_t st, const float* x_in = nullptr, int in_act = 0);
// conv_stem.hip: the 7x7 stride-2 encoder stems (2..6 input channels) on a patch-staged MFMA kernel
bool stem7_fwd_ok(const fd_conv_desc* d);
int stem7_fwd_launch(const fd_conv_desc* d, const float* x, const float* w, const float* bias, float* y, hipStream_t st);
bool wino_wgrad_ok(const fd_conv_desc* d);
long wino_wgrad_ws_floats(const fd_conv_desc* d);
int wino_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate, hipStream_t st);
// gw[m][c][t] (+)= sum_z slabs[z][m][t][c]
int fast_wgrad_finish_launch(const float* slabs, float* gw, int M, int C, int T, int splits, int accumulate, hipStream_t st);

// conv_limb.hip: 1x1 stride-1 convolutions as split-precision (3 x bf16 limbs, six products, fp32 accumulate) GEMMs on the bf16 MFMA
bool limb_fwd_ok(const fd_conv_desc* d);
bool limb_dgrad_ok(const fd_conv_desc* d);
bool limb_wgrad_ok(const fd_conv_desc* d);
long limb_wt_floats(long M, long K);                       // the pre-split weight image (conv_limb.h), in floats
long limb_gemm_ws_floats(int M, int K, int Nb, int HW);    // split-K slabs of the forward / data-gradient GEMM
int limb_weight_split_launch(const float* w, float* wt, int M, int K, int transposed, hipStream_t st);
int limb_gemm_launch(const float* wt, const float* x, float* y, const float* bias, const float* add, float* ws, int M, int K, int Nb, int HW,
                     int act, hipStream_t st);
long limb_wgrad_ws_floats(int M, int C, int Nb, int HW);
int limb_wgrad_launch(const float* x, const float* gy, float* gw, float* ws, int M, int C, int Nb, int HW, int accumulate, hipStream_t st);
// ... and the direct convolutions Winograd cannot take (stride 2) as implicit GEMMs on the same arithmetic (k_conv_limb): a problem in
// FastGemmArgs terms whose A is the pre-split image of [M][(tap, channel)] (limb_wt_floats(M, T * C) floats)
bool limb_conv_problem_ok(int M, int C, int pad_mode, int act);
int limb_conv_weight_split_launch(const float* w, float* wt, int Co, int Ci, int KH, int KW, int TA, int TB, int kh0, int dkh, int kw0, int dkw,
                                  int mode, hipStream_t st);
long limb_conv_ws_floats(const FastGemmArgs& a);
int limb_conv_launch(const FastGemmArgs& a, hipStream_t st);
int limb_conv_group_launch(const FastGemmArgs& a, const FastGemmGroup& q, hipStream_t st);
// ... and their weight gradient (3x3, stride 2, pad 1, zero padding: k_wgrad_limb_s2)
bool limb_wgrad_s2_shape_ok(int M, int C, int Hi, int Wi, int NY, int NX);
long limb_wgrad_s2_ws_floats(int M, int C, int Nb, int plane, int ntaps);
int limb_wgrad_s2_launch(const float* x, const float* gy, float* gw, float* ws, int M, int C, int Nb, int Hi, int Wi, int NY, int NX, int ntaps,
                         int accumulate, hipStream_t st);
bool limb_conv_1x1_worth(int M, long Np);
