// The Adam update of ONE element (torch.optim.Adam's single-tensor arithmetic, reference experiment.py:118-120), written
// with explicit fused / unfused operations: every kernel that applies it -- adam.hip (the multi-tensor launch) and
// xgmi.hip (the all-reduce fused with the update) -- must round identically, and the compiler's own contraction of
// `b1 * m + (1 - b1) * g` depends on the code around it.  The choices below are the ones hipcc made for the float4 path of
// adam_multi_kernel up to round 5 (so that kernel's results are unchanged):
//   m = fma(b1, m, (1 - b1) g);  v = fma(g, (1 - b2) g, b2 v);  p -= (step m) / fma(sqrt(v), rs, eps)
// with step = lr / (1 - b1^t), rs = 1 / sqrt(1 - b2^t).
#pragma once

__device__ __forceinline__ void vf_adam_update(float& p, float g, float& m, float& v, float b1, float b2, float omb1,
                                               float omb2, float step, float rs, float eps) {
#pragma clang fp contract(off)
    m = __builtin_fmaf(b1, m, omb1 * g);
    v = __builtin_fmaf(g, omb2 * g, b2 * v);
    const float den = __builtin_fmaf(sqrtf(v), rs, eps);
    p = p - (step * m) / den;
}
