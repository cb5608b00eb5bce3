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

// Packed fp32 (two lanes of a 64-bit register pair per instruction).  Written as instructions: the compiler's cost
// model splits <2 x float> arithmetic with swizzles back into scalar ops + moves, and under fp32 MFMAs every vector
// instruction counts (tools/mfma_valu.hip).
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {              // a - b
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {              // a + b
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <int K>
__device__ __forceinline__ f32x2 pk_fmak(f32x2 a, f32x2 c) {             // K a + c, K in {2, 4, 8} (inline constants)
    f32x2 d;
    if (K == 2) asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    else if (K == 4) asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a), "v"(c));
    else {
        const f32x2 a2 = pk_add(a, a);
        asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1]" : "=v"(d) : "v"(a2), "v"(c));
    }
    return d;
}
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {     // a * b + c
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_nmul4_add(f32x2 a, f32x2 c) {        // c - 4 a
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, 4.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_hi_pm_lo(f32x2 p) {                  // [p.y + p.x, p.y - p.x]
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %1 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 pk_hi_pm_2lo(f32x2 q) {                 // [q.y + 2 q.x, q.y - 2 q.x]
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[0,0,1] op_sel_hi:[0,0,1] neg_hi:[1,0,0]" : "=v"(d) : "v"(q));
    return d;
}
__device__ __forceinline__ f32x2 pk_lo_pm_hi(f32x2 a, f32x2 b) {         // [a.x + b.y, a.x - b.y]
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// Value of the lane one below within its 16-lane DPP row; lane 0 of a row gets 0 (bound_ctrl).  Through the builtin, so
// that the compiler's hazard recognizer sees a DPP instruction (a VALU write of its source needs two wait states in
// front of it).  NOT __builtin_bit_cast(int, v.w): on an ext-vector element that hands the builtin element 0
// (clang 19 / ROCm 7.2); take the element into a scalar first.
__device__ __forceinline__ float dpp_row_shr1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}

// An LDS-only workgroup barrier.  __syncthreads() is a workgroup-scope fence + barrier: the compiler drains the wave's
// outstanding GLOBAL stores in front of it (s_waitcnt vmcnt(0); loads and stores retire through one in-order counter, so
// younger loads wait too).  Where only LDS traffic has to be ordered -- the chunk loops and the LDS exchanges of the
// epilogues, whose output stores are never read back by the workgroup -- that wait is thousands of cycles per tile.
#define VF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// A pointer the program knows to be wave-uniform, as an SGPR pair (for the saddr form of global loads).
__device__ __forceinline__ const char* uniform_ptr(const void* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const char*)(((unsigned long long)hi << 32) | lo);
}
