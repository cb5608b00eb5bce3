// Fused single-head spatial self-attention forward (reference model/unet.py:258-277, the core
// between the qkv and out 1x1 convs):   O[c][i] = sum_j V[c][j] * softmax_j(Q[:,i].K[:,j] / sqrt(C))
//
// One workgroup = one view x 128 queries (4 waves x 32 queries; 64 queries / 2 waves at L=64).
// Orientation: S^T = K^T Q on v_mfma_f32_32x32x2_f32 with A = K^T (row = key), B = Q (col = query):
// every lane owns ONE query column, its keys sit in the accumulator registers.  So
//   * the row softmax is in-register (max / sum over the lane's registers + one cross-half
//     shuffle), scores never leave the register file (L <= 256 -> <= 128 accumulators / lane);
//   * the probabilities are directly the B operand of the second product O = V P^T (their
//     accumulator row order is taken as the k order, V is fetched in that order with one
//     ds_read_b128 per four MFMAs);
//   * O comes out with the query on the lane -> coalesced NCHW stores.
// K/Q are staged through LDS in 16-channel chunks, V in 32-channel tiles.
// Optionally writes P (S,L,L) for the backward pass.  Bound: fp32 MFMA.
#include "common.h"

namespace {

template <int L>
__global__ __launch_bounds__(L >= 128 ? 256 : 128) void attn_fwd_kernel(const float* __restrict__ qkv,
                                                                        float* __restrict__ out,
                                                                        float* __restrict__ P, int C, float alpha) {
    constexpr int NW = L >= 128 ? 4 : 2;
    constexpr int NTH = NW * 64;
    constexpr int QW = NW * 32;
    constexpr int NKT = L / 32;
    constexpr int CKA = 16;
    constexpr int RSV = L + 4;
    constexpr int LDS1 = CKA * L + CKA * QW, LDS3 = 32 * RSV;
    __shared__ __attribute__((aligned(16))) float lds[LDS1 > LDS3 ? LDS1 : LDS3];
    float* const Kl = lds;
    float* const Ql = lds + CKA * L;
    float* const Vl = lds;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * QW;
    const float* qb = qkv + (size_t)b * 3 * C * L;
    const float* kb = qb + (size_t)C * L;
    const float* vb = qb + (size_t)2 * C * L;

    f32x16 acc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc[t] = (f32x16){0};

    for (int c0 = 0; c0 < C; c0 += CKA) {
        __syncthreads();
        for (int e = tid; e < CKA * L / 4; e += NTH) {
            const int row = e / (L / 4), q4 = e % (L / 4);
            *reinterpret_cast<float4*>(Kl + row * L + 4 * q4) =
                *reinterpret_cast<const float4*>(kb + (size_t)(c0 + row) * L + 4 * q4);
        }
        for (int e = tid; e < CKA * QW / 4; e += NTH) {
            const int row = e / (QW / 4), q4 = e % (QW / 4);
            *reinterpret_cast<float4*>(Ql + row * QW + 4 * q4) =
                *reinterpret_cast<const float4*>(qb + (size_t)(c0 + row) * L + q0 + 4 * q4);
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < CKA / 2; ++s) {
            const float bq = Ql[(2 * s + lh) * QW + wid * 32 + li];
#pragma unroll
            for (int t = 0; t < NKT; ++t) {
                const float ak = Kl[(2 * s + lh) * L + t * 32 + li];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak, bq, acc[t], 0, 0, 0);
            }
        }
    }

    // softmax over keys: this lane's query, keys in registers (+ the other lane half)
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(alpha * (acc[t][r] - mx));
            acc[t][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= inv;

    const int qi = q0 + wid * 32 + li;
    if (P) {
        float* pr = P + ((size_t)b * L + qi) * L;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(pr + t * 32 + 8 * g + 4 * lh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }

    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
        for (int e = tid; e < 32 * L / 4; e += NTH) {
            const int row = e / (L / 4), q4 = e % (L / 4);
            *reinterpret_cast<float4*>(Vl + row * RSV + 4 * q4) =
                *reinterpret_cast<const float4*>(vb + (size_t)(c0 + row) * L + 4 * q4);
        }
        __syncthreads();
        f32x16 o = {0};
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 av = *reinterpret_cast<const float4*>(Vl + li * RSV + t * 32 + 8 * g + 4 * lh);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, acc[t][4 * g + 0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, acc[t][4 * g + 1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, acc[t][4 * g + 2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, acc[t][4 * g + 3], o, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((size_t)b * C + c) * L + qi] = o[r];
        }
    }
}

}  // namespace

extern "C" {

// qkv [S][3C][L] (q | k | v channel thirds), out [S][C][L], P [S][L][L] or NULL.
// L in {64, 256, 1024(no)}: spatial 8x8 or 16x16; C multiple of 32.
int vf_attention_fwd(const float* qkv, float* out, float* P, int S, int C, int L, void* stream) {
    if (S <= 0) return 0;
    if (C % 32 != 0) return (int)hipErrorInvalidValue;
    const float alpha = 1.0f / sqrtf((float)C);
    hipStream_t st = (hipStream_t)stream;
    if (L == 256)
        hipLaunchKernelGGL(attn_fwd_kernel<256>, dim3(2, S), dim3(256), 0, st, qkv, out, P, C, alpha);
    else if (L == 64)
        hipLaunchKernelGGL(attn_fwd_kernel<64>, dim3(1, S), dim3(128), 0, st, qkv, out, P, C, alpha);
    else
        return (int)hipErrorInvalidValue;
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
