// Shared device helpers for the ViewFusion gfx950 (CDNA4) kernels.
// Wavefront = 64 lanes everywhere in this tree.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VF_WAVE 64

// Every launcher: enqueue on `stream`, never sync / allocate, return hipError_t as int.
#define VF_RETURN_LAST_ERROR() return (int)hipGetLastError()

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, VF_WAVE);
    return v;
}

// sum over aligned segments of SEG lanes (SEG a power of two <= 64); every lane of a segment gets the sum
template <int SEG>
__device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
    for (int o = SEG / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, VF_WAVE);
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, VF_WAVE));
    return v;
}

// Sum over a whole workgroup of NT threads (NT multiple of 64); result broadcast to all
// threads.  `red` is an LDS array of >= NT/64 floats.  Deterministic (fixed tree).
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` from the previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}

// sigmoid(z) = 1 / (1 + 2^(-z log2 e)) as v_exp_f32 + v_rcp_f32 (1 ulp each): expf() + an IEEE division cost ~25 VALU
// instructions per element, which made the "HBM-bound" GroupNorm+Swish kernels VALU-bound.  Saturates correctly:
// z -> -inf gives 2^(+inf) = inf, rcp(inf) = 0; z -> +inf gives rcp(1) = 1.
__device__ __forceinline__ float sigmoid_f(float z) {
#ifdef VF_EXACT_SILU
    return 1.0f / (1.0f + expf(-z));
#else
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896341f));
#endif
}
__device__ __forceinline__ float silu_f(float z) { return z * sigmoid_f(z); }

// XCD-aware remap of a 1-D grid: consecutive *logical* ids run on the same XCD (blocks are
// dealt round-robin over the 8 XCDs), so neighbouring tiles share one L2.  Bijective for any
// grid size.  Speed only -- never correctness.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u;
    const unsigned xcd = bid & 7u, slot = bid >> 3;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + slot;
}
