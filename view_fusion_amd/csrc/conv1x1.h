// Internal interface between conv.hip (dispatch of vf_conv_fwd / vf_conv1x1_cat_*) and conv1x1.hip (the training-size
// 1x1 kernel).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>

struct C11Args {
    const float* x;
    const float* x2;      // second source of the channel concatenation [x | x2], or null
    const float* w;       // packed [co tile 64][ci chunk 32][group 4][co 64][ci 8]
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    float* y2;            // second destination of the split output [y | y2], or null
    int S, Cin, Cout, C1, C1o, HW, hwsh, npx, nct;     // npx = S * HW pixels, nct = 64-channel output tiles
};

bool vfi_conv1x1_supported(const C11Args& a);
int vfi_conv1x1_halves(const C11Args& a);      // 2: 128-channel workgroups, 1: 64-channel ones, 0: not taken
int vfi_conv1x1_launch(const C11Args& a, hipStream_t st);
