// Internal interface between conv.hip and conv_narrow.hip (narrow-channel 3x3 weight gradient).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/fdhip.h"

struct NarrowWgradArgs {
    const float* X; const float* dY; float* slabs;   // slabs [workgroups][M][C][9]
    int N, M, H, W, pad_mode;
};

bool narrow_wgrad_ok(const fd_conv_desc* d);
long narrow_wgrad_ws_floats(const fd_conv_desc* d);
int narrow_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate,
                        hipStream_t st);

bool stem_wgrad_ok(const fd_conv_desc* d);
long stem_wgrad_ws_floats(const fd_conv_desc* d);
int stem_wgrad_launch(const fd_conv_desc* d, const float* x, const float* gy, float* gw, float* ws, int accumulate, hipStream_t st);
