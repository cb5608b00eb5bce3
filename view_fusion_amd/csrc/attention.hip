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

    // K / Q chunk staging: all loads of a chunk are issued together into registers, one chunk ahead of their use
    // (a `*lds = *global` loop compiles to load -> wait -> store per element: six serial round trips per chunk)
    constexpr int NK4 = CKA * L / 4 / NTH, NQ4 = CKA * QW / 4 / NTH;
    static_assert(NK4 * NTH * 4 == CKA * L && NQ4 * NTH * 4 == CKA * QW, "whole passes");
    // (named registers: the array form of these staging sets is not promoted out of scratch memory)
    float4 kr0, kr1, kr2, kr3, qr0, qr1;
    kr0 = kr1 = kr2 = kr3 = qr0 = qr1 = make_float4(0.f, 0.f, 0.f, 0.f);
    static_assert(NK4 <= 4 && NQ4 <= 2, "staging register sets");
#define VF_AT_LK(I, C0) if constexpr ((I) < NK4) { const int e = tid + (I) * NTH;                          \
        kr##I = *reinterpret_cast<const float4*>(kb + (size_t)((C0) + e / (L / 4)) * L + 4 * (e % (L / 4))); }
#define VF_AT_LQ(I, C0) if constexpr ((I) < NQ4) { const int e = tid + (I) * NTH;                          \
        qr##I = *reinterpret_cast<const float4*>(qb + (size_t)((C0) + e / (QW / 4)) * L + q0 + 4 * (e % (QW / 4))); }
#define VF_AT_SK(I) if constexpr ((I) < NK4) { const int e = tid + (I) * NTH;                               \
        *reinterpret_cast<float4*>(Kl + (e / (L / 4)) * L + 4 * (e % (L / 4))) = kr##I; }
#define VF_AT_SQ(I) if constexpr ((I) < NQ4) { const int e = tid + (I) * NTH;                               \
        *reinterpret_cast<float4*>(Ql + (e / (QW / 4)) * QW + 4 * (e % (QW / 4))) = qr##I; }
#define VF_AT_LOAD(C0) { VF_AT_LK(0, C0) VF_AT_LK(1, C0) VF_AT_LK(2, C0) VF_AT_LK(3, C0) VF_AT_LQ(0, C0) VF_AT_LQ(1, C0) }
    VF_AT_LOAD(0);
    for (int c0 = 0; c0 < C; c0 += CKA) {
        __syncthreads();
        VF_AT_SK(0) VF_AT_SK(1) VF_AT_SK(2) VF_AT_SK(3) VF_AT_SQ(0) VF_AT_SQ(1)
        __syncthreads();
        VF_AT_LOAD(min(c0 + CKA, C - CKA));              // next chunk (clamped: the last one is re-read, unused)
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
#undef VF_AT_LOAD
#undef VF_AT_LK
#undef VF_AT_LQ
#undef VF_AT_SK
#undef VF_AT_SQ
    // first V chunk: requested now, lands while the softmax is computed
    constexpr int NV4 = 32 * L / 4 / NTH;
    static_assert(NV4 * NTH * 4 == 32 * L, "whole passes");
    static_assert(NV4 <= 8, "staging register set");
    float4 vr0, vr1, vr2, vr3, vr4, vr5, vr6, vr7;
    vr0 = vr1 = vr2 = vr3 = vr4 = vr5 = vr6 = vr7 = make_float4(0.f, 0.f, 0.f, 0.f);
#define VF_AT_LV(I, C0) if constexpr ((I) < NV4) { const int e = tid + (I) * NTH;                          \
        vr##I = *reinterpret_cast<const float4*>(vb + (size_t)((C0) + e / (L / 4)) * L + 4 * (e % (L / 4))); }
#define VF_AT_SV(I) if constexpr ((I) < NV4) { const int e = tid + (I) * NTH;                               \
        *reinterpret_cast<float4*>(Vl + (e / (L / 4)) * RSV + 4 * (e % (L / 4))) = vr##I; }
#define VF_AT_LOADV(C0) { VF_AT_LV(0, C0) VF_AT_LV(1, C0) VF_AT_LV(2, C0) VF_AT_LV(3, C0) VF_AT_LV(4, C0) VF_AT_LV(5, C0) VF_AT_LV(6, C0) VF_AT_LV(7, C0) }
    VF_AT_LOADV(0);

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
        VF_AT_SV(0) VF_AT_SV(1) VF_AT_SV(2) VF_AT_SV(3) VF_AT_SV(4) VF_AT_SV(5) VF_AT_SV(6) VF_AT_SV(7)
        __syncthreads();
        VF_AT_LOADV(min(c0 + 32, C - 32));               // next chunk, issued before this chunk's output stores
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
#undef VF_AT_LOADV
#undef VF_AT_LV
#undef VF_AT_SV
}


// Key-split variant: one workgroup = one view x 32 queries, wave w owns keys [32w, 32w+32) -- L/32 waves.
// The 47 us dependent MFMA chain of one wave in the kernel above (all L keys x C channels for its 32 queries) is cut
// L/32-fold, and the grid grows 4x (S x L/32 workgroups): the sampler (S = 1..12 views) no longer runs its seven
// attention layers on a dozen waves, and at S = 96 every CU holds several workgroups whose phases overlap.
//   1. S^T tile (32 keys x 32 queries) per wave, K/Q staged through LDS in 32-channel chunks;
//   2. softmax across waves: per-wave max / sum of the lane's query through LDS (two barriers), fixed order;
//   3. P (normalised) -> LDS [key][query] (aliases the K/Q staging area) and, optionally, global (S,L,L);
//   4. O = V P^T: 32-channel tiles dealt round-robin to the waves; V comes straight from global memory in the
//      k order of the accumulator rows (one float4 per four MFMAs), P^T from LDS.
template <int L>
__global__ __launch_bounds__(L / 32 * 64) void attn_fwd_split_kernel(const float* __restrict__ qkv,
                                                                     float* __restrict__ out, float* __restrict__ P,
                                                                     int C, float alpha) {
    constexpr int NW = L / 32, NTH = NW * 64, CKA = 32;
    constexpr int KS = L + 32;                       // K row stride: the two lane halves hit different banks
    constexpr int PS = 40;                           // P row stride, same reason
    constexpr int STAGE = CKA * KS + CKA * 32, PSZ = L * PS;
    __shared__ __attribute__((aligned(16))) float lds[STAGE > PSZ ? STAGE : PSZ];
    __shared__ float red[2][NW][32];
    float* const Kl = lds;
    float* const Ql = lds + CKA * KS;
    float* const Pl = lds;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int b = blockIdx.y, q0 = blockIdx.x * 32;
    const float* qb = qkv + (size_t)b * 3 * C * L;
    const float* kb = qb + (size_t)C * L;
    const float* vb = qb + (size_t)2 * C * L;

    // ---- 1. scores ----
    constexpr int NK4 = CKA * L / 4 / NTH;           // = 4 float4 per thread
    static_assert(NK4 == 4, "K staging");
    // Q chunk = 32 channels x 32 queries = 256 float4: L = 256 -> every thread loads one (waves >= 4 repeat it),
    // L = 64 (128 threads) -> two per thread
    constexpr int NQ4 = NTH >= 256 ? 1 : 256 / NTH;
    const int eq = tid & 255;
    // (named registers + macros: staging arrays filled behind a lambda are not promoted out of scratch memory)
    float4 kr0, kr1, kr2, kr3, qr, qr1;
    qr1 = make_float4(0.f, 0.f, 0.f, 0.f);
#define VF_AS_LK(I, C0) { const int e = tid + (I) * NTH;                                                   \
        kr##I = *reinterpret_cast<const float4*>(kb + (size_t)((C0) + e / (L / 4)) * L + 4 * (e % (L / 4))); }
#define VF_AS_SK(I) { const int e = tid + (I) * NTH;                                                        \
        *reinterpret_cast<float4*>(Kl + (e / (L / 4)) * KS + 4 * (e % (L / 4))) = kr##I; }
#define load_chunk(C0) { VF_AS_LK(0, C0) VF_AS_LK(1, C0) VF_AS_LK(2, C0) VF_AS_LK(3, C0)                  \
        qr = *reinterpret_cast<const float4*>(qb + (size_t)((C0) + eq / 8) * L + q0 + 4 * (eq % 8));           \
        if constexpr (NQ4 == 2)                                                                                \
            qr1 = *reinterpret_cast<const float4*>(qb + (size_t)((C0) + (eq + NTH) / 8) * L + q0 + 4 * (eq % 8)); }
    f32x16 acc = {0};
    load_chunk(0);
    for (int c0 = 0; c0 < C; c0 += CKA) {
        __syncthreads();
        VF_AS_SK(0) VF_AS_SK(1) VF_AS_SK(2) VF_AS_SK(3)
        if (tid < 256) *reinterpret_cast<float4*>(Ql + (eq / 8) * 32 + 4 * (eq % 8)) = qr;
        if constexpr (NQ4 == 2) *reinterpret_cast<float4*>(Ql + ((eq + NTH) / 8) * 32 + 4 * (eq % 8)) = qr1;
        __syncthreads();
        load_chunk(min(c0 + CKA, C - CKA));          // next chunk (clamped: the last one is re-read, unused)
#pragma unroll
        for (int s = 0; s < CKA / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kl[(2 * s + lh) * KS + wid * 32 + li],
                                                       Ql[(2 * s + lh) * 32 + li], acc, 0, 0, 0);
    }

    // first V batch of this wave's first channel tile: requested now, lands during the softmax
    constexpr int NB = L / 64;                       // batches of 64 keys = 8 float4 per lane
    const int ntile = C / 32;
#undef load_chunk
#undef VF_AS_LK
#undef VF_AS_SK
    float4 va0, va1, va2, va3, va4, va5, va6, va7, vn0, vn1, vn2, vn3, vn4, vn5, vn6, vn7;
#define VF_AS_LV1(V, G, SRC) V##G = *reinterpret_cast<const float4*>((SRC) + 8 * (G));
#define load_v(V, TILE, KB) { const float* src_ = vb + (size_t)((TILE) * 32 + li) * L + (KB) * 64 + 4 * lh;  \
        VF_AS_LV1(V, 0, src_) VF_AS_LV1(V, 1, src_) VF_AS_LV1(V, 2, src_) VF_AS_LV1(V, 3, src_)              \
        VF_AS_LV1(V, 4, src_) VF_AS_LV1(V, 5, src_) VF_AS_LV1(V, 6, src_) VF_AS_LV1(V, 7, src_) }
    load_v(va, min(wid, ntile - 1), 0);

    // ---- 2. softmax over all keys of the lane's query ----
    float mx = acc[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, acc[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lh == 0) red[0][wid][li] = mx;
    __syncthreads();                                 // (also: every wave is done with Kl / Ql)
    mx = red[0][0][li];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[0][w][li]);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[r] = expf(alpha * (acc[r] - mx));
        sum += acc[r];
    }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) red[1][wid][li] = sum;
    __syncthreads();
    sum = red[1][0][li];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[1][w][li];
    const float inv = 1.0f / sum;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= inv;

    // ---- 3. P -> LDS [key][query] (+ global) ----
#pragma unroll
    for (int r = 0; r < 16; ++r) Pl[(wid * 32 + 8 * (r >> 2) + 4 * lh + (r & 3)) * PS + li] = acc[r];
    if (P) {
        float* pr = P + ((size_t)b * L + q0 + li) * L + wid * 32 + 4 * lh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4*>(pr + 8 * g) = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    }
    __syncthreads();

    // ---- 4. O tile = V[tile] P^T ----
    for (int t = wid; t < ntile; t += NW) {
        f32x16 o = {0};
#pragma unroll 1
        for (int kbt = 0; kbt < NB; ++kbt) {
            // next batch: the following keys of this tile, or the first keys of this wave's next tile (clamped)
            const int nt = kbt + 1 < NB ? t : min(t + NW, ntile - 1), nk = kbt + 1 < NB ? kbt + 1 : 0;
            load_v(vn, nt, nk);
            const float* pl = Pl + (kbt * 64 + 4 * lh) * PS + li;
#define VF_AS_PV(G) {                                                                                        \
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(va##G.x, pl[(8 * (G) + 0) * PS], o, 0, 0, 0);          \
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(va##G.y, pl[(8 * (G) + 1) * PS], o, 0, 0, 0);          \
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(va##G.z, pl[(8 * (G) + 2) * PS], o, 0, 0, 0);          \
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(va##G.w, pl[(8 * (G) + 3) * PS], o, 0, 0, 0); }
            VF_AS_PV(0) VF_AS_PV(1) VF_AS_PV(2) VF_AS_PV(3) VF_AS_PV(4) VF_AS_PV(5) VF_AS_PV(6) VF_AS_PV(7)
#undef VF_AS_PV
            va0 = vn0; va1 = vn1; va2 = vn2; va3 = vn3; va4 = vn4; va5 = vn5; va6 = vn6; va7 = vn7;
        }
        float* ob = out + ((size_t)b * C + t * 32 + 4 * lh) * L + q0 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(size_t)((r & 3) + 8 * (r >> 2)) * L] = o[r];
    }
#undef load_v
#undef VF_AS_LV1
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
    // few views (sampler): the key-split kernel cuts the per-wave MFMA chain 8-fold; many views (training): it
    // would re-stream K and V once per 32 queries (L2-bound), the 128-query kernel wins from S ~ 50 on
    if (L == 256 && S <= 32)
        hipLaunchKernelGGL(attn_fwd_split_kernel<256>, dim3(8, S), dim3(512), 0, st, qkv, out, P, C, alpha);
    else if (L == 64)
        hipLaunchKernelGGL(attn_fwd_split_kernel<64>, dim3(2, S), dim3(128), 0, st, qkv, out, P, C, alpha);
    else if (L == 256)
        hipLaunchKernelGGL(attn_fwd_kernel<256>, dim3(2, S), dim3(256), 0, st, qkv, out, P, C, alpha);
    else
        return (int)hipErrorInvalidValue;
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
