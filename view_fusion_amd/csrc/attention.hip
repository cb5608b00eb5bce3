// Fused single-head spatial self-attention forward (reference model/unet.py:258-277, the core
// between the qkv and out 1x1 convs):   O[c][i] = sum_j V[c][j] * softmax_j(Q[:,i].K[:,j] / sqrt(C))
//
// Orientation (both kernels): S^T = K^T Q on v_mfma_f32_32x32x2_f32 with A = K^T (row = key), B = Q (col = query):
// every lane owns ONE query column, its keys sit in the accumulator registers.  So
//   * the row softmax is in-register (max / sum over the lane's registers + one cross-half shuffle, then the
//     statistics of the waves sharing the query block through LDS); scores never leave the register file;
//   * the probabilities are the B operand of the second product O = V P^T: the accumulator row order is taken as
//     the k order, V is fetched in that order with one 16-byte read per four MFMAs;
//   * O comes out with the query on the lane -> coalesced NCHW stores.
// Optionally writes P (S,L,L) for the backward pass.  Bound: fp32 MFMA.
//   attn_fwd_q32_kernel       L = 256, S >= 17 (training, round 5): 32 queries per workgroup, 4 waves = 4 key quarters,
//                             three workgroups per compute unit, the eight workgroups of a view on one XCD
//   attn_fwd_kh_kernel        L = 256 where 2 S workgroups fill the chip better than 8 S (S = 97..128, 225..256, ...):
//                             128 queries per workgroup, 8 waves = 4 query blocks x 2 key halves; K/Q chunks and V
//                             tiles double-buffered in LDS
//   attn_fwd_split_kernel<L>  L = 64: 32 queries per workgroup, one wave per 32 keys
//   attn_fwd_q16_kernel<L>    S <= 16 (the sampler): 16 queries per workgroup
#include "common.h"

namespace {

// Many views (training), L = 256: one workgroup = one view x 128 queries = 4 query blocks of 32, each shared by
// TWO waves that own half of the keys -- 8 waves, two per SIMD, so a SIMD interleaves two 23 us MFMA chains (one
// wave per query block with all 256 keys, 4 waves per workgroup, measured 73 us per launch at S = 96; this
// kernel 66 us).  K and V are streamed once per 128 queries (2 x S workgroups).  K/Q chunks and V tiles are double-buffered in LDS (ONE barrier per chunk / tile: the next
// one is written while this one feeds the MFMAs).  Softmax statistics of the two key halves are combined through
// LDS; the partial O tiles of the upper half are handed to the lower-half wave through a double-buffered LDS
// slot, one channel tile behind the MFMAs.
__global__ __launch_bounds__(512) void attn_fwd_kh_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                          float* __restrict__ P, int C, float alpha) {
    constexpr int L = 256, NTH = 512, QW = 128, NKT = 4, CKA = 16, RSV = L + 4;
    constexpr int KS = L + 32;                       // K row stride: the two lane halves hit different banks
    constexpr int KQ = CKA * KS + CKA * QW, VT = 32 * RSV;
    __shared__ __attribute__((aligned(16))) float lds[2 * (KQ > VT ? KQ : VT)];
    __shared__ float oex[2][4][16][64];
    __shared__ float red[2][2][QW];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int qb = wid & 3, kh = wid >> 2;           // waves w and w+4 (same SIMD) share a query block
    const int b = blockIdx.y, q0 = blockIdx.x * QW;
    const float* qp = qkv + (size_t)b * 3 * C * L;
    const float* kp = qp + (size_t)C * L;
    const float* vp = qp + (size_t)2 * C * L;

#ifdef VF_ATTN_STAMPS   // diagnostic build only (tools/attn_stamps.py): phase clocks of wave 0, in the P slot
    long long st_[6] = {clock64(), 0, 0, 0, 0, 0}, rt0_ = wall_clock64();
#endif
    f32x16 acc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc[t] = (f32x16){0};

    // K chunk = 16 x 256 floats = 1024 float4 (2 per thread), Q chunk = 16 x 128 = 512 float4 (1 per thread);
    // (32-channel chunks, i.e. half the barriers, measured 2 % slower)
    float4 kr0, kr1, qr;
#define VF_AK_LOAD(C0) {                                                                                    \
        kr0 = *reinterpret_cast<const float4*>(kp + (size_t)((C0) + tid / 64) * L + 4 * (tid % 64));          \
        kr1 = *reinterpret_cast<const float4*>(kp + (size_t)((C0) + 8 + tid / 64) * L + 4 * (tid % 64));      \
        qr = *reinterpret_cast<const float4*>(qp + (size_t)((C0) + tid / 32) * L + q0 + 4 * (tid % 32)); }
#define VF_AK_STORE(BUF) { float* kl_ = lds + (BUF) * KQ;                                                    \
        *reinterpret_cast<float4*>(kl_ + (tid / 64) * KS + 4 * (tid % 64)) = kr0;                             \
        *reinterpret_cast<float4*>(kl_ + (8 + tid / 64) * KS + 4 * (tid % 64)) = kr1;                         \
        *reinterpret_cast<float4*>(kl_ + CKA * KS + (tid / 32) * QW + 4 * (tid % 32)) = qr; }
    VF_AK_LOAD(0);
    VF_AK_STORE(0);
    VF_AK_LOAD(min(CKA, C - CKA));
    __syncthreads();
    for (int c0 = 0, cur = 0; c0 < C; c0 += CKA, cur ^= 1) {
        VF_AK_STORE(cur ^ 1);                            // next chunk (its buffer was last read before the barrier)
        VF_AK_LOAD(min(c0 + 2 * CKA, C - CKA));          // the one after (clamped: re-reads the last one, unused)
        const float* Kl = lds + cur * KQ;
        const float* Ql = Kl + CKA * KS;
        // operand fragments of step s + 1 are read from LDS before the MFMAs of step s are issued
        const float* kq = Kl + lh * KS + kh * NKT * 32 + li;
        const float* qq = Ql + lh * QW + qb * 32 + li;
        float bq = qq[0], ak0 = kq[0], ak1 = kq[32], ak2 = kq[64], ak3 = kq[96];
#pragma unroll
        for (int s = 0; s < CKA / 2; ++s) {
            float bqn = bq, an0 = ak0, an1 = ak1, an2 = ak2, an3 = ak3;
            if (s + 1 < CKA / 2) {
                bqn = qq[(2 * s + 2) * QW];
                an0 = kq[(2 * s + 2) * KS];
                an1 = kq[(2 * s + 2) * KS + 32];
                an2 = kq[(2 * s + 2) * KS + 64];
                an3 = kq[(2 * s + 2) * KS + 96];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak0, bq, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak1, bq, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak2, bq, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak3, bq, acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            bq = bqn; ak0 = an0; ak1 = an1; ak2 = an2; ak3 = an3;
        }
        __syncthreads();
    }
#undef VF_AK_LOAD
#undef VF_AK_STORE
#ifdef VF_ATTN_STAMPS
    st_[1] = clock64();
#endif
    // V tiles (32 channels x 256 keys = 2048 float4, 4 per thread): tile 0 is requested now and lands during the
    // softmax
    float4 vr0, vr1, vr2, vr3;
#define VF_AK_LV(I, C0) { const int e = tid + (I) * NTH;                                                   \
        vr##I = *reinterpret_cast<const float4*>(vp + (size_t)((C0) + e / 64) * L + 4 * (e % 64)); }
#define VF_AK_SV(I, BUF) { const int e = tid + (I) * NTH;                                                   \
        *reinterpret_cast<float4*>(lds + (BUF) * VT + (e / 64) * RSV + 4 * (e % 64)) = vr##I; }
#define VF_AK_LOADV(C0) { VF_AK_LV(0, C0) VF_AK_LV(1, C0) VF_AK_LV(2, C0) VF_AK_LV(3, C0) }
#define VF_AK_STOREV(BUF) { VF_AK_SV(0, BUF) VF_AK_SV(1, BUF) VF_AK_SV(2, BUF) VF_AK_SV(3, BUF) }
    VF_AK_LOADV(0);

    // softmax over all keys of the lane's query: this wave's 128 keys, then the partner wave's statistics
    const int ql = qb * 32 + li;
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lh == 0) red[0][kh][ql] = mx;
    __syncthreads();
    mx = fmaxf(red[0][0][ql], red[0][1][ql]);
    // exp(alpha (s - max)) as ONE v_exp_f32 per score: 2^(c s - c max), c = alpha log2(e); the argument is <= 0, so
    // the range handling of expf() has nothing to do
    const float c2 = alpha * 1.44269504088896341f, mc = mx * c2;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(acc[t][r], c2, -mc));
            acc[t][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) red[1][kh][ql] = sum;
    VF_AK_STOREV(0);                                     // (the K/Q buffers are dead since the loop's last barrier)
    VF_AK_LOADV(min(32, C - 32));
    __syncthreads();
    const float inv = 1.0f / (red[1][0][ql] + red[1][1][ql]);
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= inv;

    const int qi = q0 + ql;
#ifdef VF_ATTN_STAMPS
    st_[2] = clock64();
    if (false) {
#else
    if (P) {
#endif
        float* pr = P + ((size_t)b * L + qi) * L + kh * NKT * 32;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(pr + t * 32 + 8 * g + 4 * lh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
#ifdef VF_ATTN_STAMPS
    st_[3] = clock64();
#endif

    // O = V P^T, one 32-channel tile per iteration: tile c0 feeds the MFMAs from V buffer `cur` while tile c0 + 32
    // is written to the other buffer; the upper-half wave's partial tile goes to oex[cur] and is added and stored
    // by its partner at the top of the next iteration
    f32x16 oprev = {0};
    for (int c0 = 0, cur = 0; c0 < C; c0 += 32, cur ^= 1) {
        if (kh == 0 && c0 > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 - 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[((size_t)b * C + c) * L + qi] = oprev[r] + oex[cur ^ 1][qb][r][lane];
            }
        }
        VF_AK_STOREV(cur ^ 1);                           // tile c0 + 32
        VF_AK_LOADV(min(c0 + 64, C - 32));               // tile c0 + 64 (clamped)
        const float* Vl = lds + cur * VT;
        f32x16 o = {0};
        const float* vq = Vl + li * RSV + kh * NKT * 32 + 4 * lh;
        float4 av = *reinterpret_cast<const float4*>(vq);
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 avn = av;
                if (t * 4 + g + 1 < NKT * 4) avn = *reinterpret_cast<const float4*>(vq + (t * 4 + g + 1) * 8);
                __builtin_amdgcn_sched_barrier(0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, acc[t][4 * g + 0], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, acc[t][4 * g + 1], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, acc[t][4 * g + 2], o, 0, 0, 0);
                o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, acc[t][4 * g + 3], o, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                av = avn;
            }
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) oex[cur][qb][r][lane] = o[r];
        } else {
            oprev = o;
        }
        // (LDS-only barrier: __syncthreads() would also drain this wave's global stores -- the P rows written above and
        // the output rows of the previous tile -- at every one of the C / 32 iterations)
        VF_LDS_BARRIER();
    }
    if (kh == 0) {                                       // last tile (its oex slot is (C / 32 - 1) & 1)
        const int last = (C / 32 - 1) & 1;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = C - 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((size_t)b * C + c) * L + qi] = oprev[r] + oex[last][qb][r][lane];
        }
    }
#undef VF_AK_LOADV
#undef VF_AK_STOREV
#undef VF_AK_LV
#undef VF_AK_SV
#ifdef VF_ATTN_STAMPS
    if (tid == 0 && P) {
        long long* o = reinterpret_cast<long long*>(P) + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8;
        o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = clock64(); o[5] = 0;
        o[6] = rt0_; o[7] = wall_clock64();
    }
#endif
}

// Many views, L = 256, 32 queries per workgroup (round 5).  The 128-query kernel above launches 2 S workgroups: at S = 96
// that is 192 workgroups on 256 compute units -- a quarter of the chip idles whatever the kernel does inside.  This one
// launches 8 S workgroups of FOUR waves (three resident per compute unit: 768 = 3 x 256 at S = 96): wave w owns the
// keys [64 w, 64 w + 64) of the workgroup's 32 queries.
//   * scores: as above (K / Q chunks of 16 channels double-buffered in LDS, one barrier per chunk), two key tiles per wave;
//   * softmax: statistics of the four key quarters combined through LDS;
//   * O = V P^T: a wave needs only ITS 64 key columns of V, so every wave stages its own [32 channels][32 keys] pieces
//     (coalesced 128-byte row segments -> a wave-private LDS slot -> the MFMA k order; no workgroup barrier), and the four
//     partial O tiles are summed through LDS once per 32-channel tile: every wave finishes four of the sixteen rows;
//   * K and V are streamed once per 32 queries, i.e. four times as often as above: the eight workgroups of a view are
//     placed on ONE XCD (block id -> (view, query block) below), so the re-reads hit that XCD's L2.
// DSCORE (the same kernel as the first half of the attention BACKWARD, round 5): "K" = V, "Q" = dO, so the score phase
// leaves dP^T = V^T dO in the accumulators, again one query per lane; with the forward's probabilities P (read in the layout
// the forward stored them) the softmax backward is in-register:  dS = P (dP - sum_keys P dP)  -> dS (S, L, L), which the
// dK product still needs.  The second phase then runs with K in the place of V: alpha K dS^T = dQ -> the q third of `out` =
// dqkv (S, 3C, L).  Replaces the dP product (which wrote 25 MB at S = 96), the softmax-backward launch that read it back
// and the dQ product (which read dS again).
template <bool DSCORE>
__global__ __launch_bounds__(256, 3) void attn_fwd_q32_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                              float* __restrict__ P, int C, float alpha, int S,
                                                              const float* __restrict__ dO, float* __restrict__ dS) {
    constexpr int L = 256, NTH = 256, NKT = 2, CKA = 16;
    constexpr int KS = L + 32;                       // K row stride: the two lane halves hit different banks
    constexpr int KQ = CKA * KS + CKA * 32;          // one K / Q chunk
    constexpr int RSV = 36, VW = 32 * RSV;           // a wave's V piece [32 channels][32 keys (+ 4)]
    constexpr int OEX = 4 * 16 * 64;                 // partial O tiles of the four waves
    constexpr int PV = 4 * VW + 2 * OEX;
    __shared__ __attribute__((aligned(16))) float lds[2 * KQ > PV ? 2 * KQ : PV];
    __shared__ float red[2][4][32];

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // block id -> (view, query block): blocks i, i + 8, i + 16 ... run on the same XCD; the eight query blocks of a view
    // take eight consecutive slots of one XCD (views past the last whole group of eight: plain order)
    int b, u;
    {
        const int i = blockIdx.x, nfull = (S >> 3) << 6;
        if (i < nfull) { const int j = i >> 3; b = ((j >> 3) << 3) + (i & 7); u = j & 7; }
        else { const int r = i - nfull; b = ((S >> 3) << 3) + (r >> 3); u = r & 7; }
    }
    const int q0 = u * 32;
    // operands of the two phases: scores = kp^T qp, second phase = vp (.)^T
    const float* qp = DSCORE ? dO + (size_t)b * C * L : qkv + (size_t)b * 3 * C * L;
    const float* kp = qkv + ((size_t)b * 3 + (DSCORE ? 2 : 1)) * C * L;
    const float* vp = qkv + ((size_t)b * 3 + (DSCORE ? 1 : 2)) * C * L;
#ifdef VF_ATTN_STAMPS   // diagnostic build only (tools/attn_stamps.py q32): phase clocks of wave 0, in the P slot
    long long st_[6] = {clock64(), 0, 0, 0, 0, 0}, rt0_ = wall_clock64();
#endif

    f32x16 acc[NKT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) acc[t] = (f32x16){0};

    // K chunk = 16 x 256 floats = 1024 float4 (wave w: rows 4 w .. 4 w + 3, one row per instruction); Q chunk = 16 x 32 =
    // 128 float4 (waves 2 and 3 repeat what 0 and 1 write: no branch in the loop).  In-kernel clocks of the first version:
    //   * a lone workgroup spent 1900 cycles per chunk for 1024 cycles of MFMAs -- and still did with the chunks requested
    //     two iterations ahead: not load latency, but what stands between the barrier and the first MFMA (five LDS stores,
    //     twenty address instructions, five loads, then the first operand reads queue up behind the stores);
    //   * so: the first operands are read right behind the barrier, the stores and loads of the staging sit BEHIND the
    //     first pair of MFMAs, and the loads take a scalar base + one per-lane offset (no vector address arithmetic).
    // Chunks are requested two iterations ahead (two register sets, the loop runs in pairs).
    f32x4 ka0, ka1, ka2, ka3, qa, kb0, kb1, kb2, kb3, qb_;
    const int qe = tid & 127;
    unsigned kvo = (unsigned)((4 * w * L + 4 * lane) * 4);                      // bytes
    unsigned qvo = (unsigned)(((qe >> 3) * L + q0 + 4 * (qe & 7)) * 4);
    float* const kst = lds + 4 * w * KS + 4 * lane;
    float* const qst = lds + CKA * KS + (qe >> 3) * 32 + 4 * (qe & 7);
#define VF_GLD4(BASE, OFS, IMM) (*(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) char*)(BASE) + (OFS) + (IMM)))
#define VF_A3_LK(R, I, KC) R##I = VF_GLD4(KC, kvo, (I) * (L * 4));
#define VF_A3_LOAD(R, Q, C0) { const char* kc_ = uniform_ptr(kp + (size_t)(C0) * L);                               \
        const char* qc_ = uniform_ptr(qp + (size_t)(C0) * L);                                                    \
        asm("" : "+s"(kc_), "+s"(qc_), "+v"(kvo), "+v"(qvo));                                                    \
        VF_A3_LK(R, 0, kc_) VF_A3_LK(R, 1, kc_) VF_A3_LK(R, 2, kc_) VF_A3_LK(R, 3, kc_)                             \
        Q = VF_GLD4(qc_, qvo, 0); }
#define VF_A3_SK(R, I, BUF) *reinterpret_cast<f32x4*>(kst + (BUF) * KQ + (I) * KS) = R##I;
#define VF_A3_STORE(R, Q, BUF) { VF_A3_SK(R, 0, BUF) VF_A3_SK(R, 1, BUF) VF_A3_SK(R, 2, BUF) VF_A3_SK(R, 3, BUF)   \
        *reinterpret_cast<f32x4*>(qst + (BUF) * KQ) = Q; }
#define VF_A3_MFMAS(BUF, MID) {                                                                                 \
        const float* Kl = lds + (BUF) * KQ;                                                                    \
        const float* kq = Kl + lh * KS + w * (NKT * 32) + li;                                                  \
        const float* qq = Kl + CKA * KS + lh * 32 + li;                                                        \
        float bq = qq[0], ak0 = kq[0], ak1 = kq[32];                                                           \
        _Pragma("unroll")                                                                                      \
        for (int s = 0; s < CKA / 2; ++s) {                                                                    \
            float bqn = bq, an0 = ak0, an1 = ak1;                                                              \
            if (s + 1 < CKA / 2) {                                                                             \
                bqn = qq[(2 * s + 2) * 32];                                                                    \
                an0 = kq[(2 * s + 2) * KS];                                                                    \
                an1 = kq[(2 * s + 2) * KS + 32];                                                               \
            }                                                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak0, bq, acc[0], 0, 0, 0);                           \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak1, bq, acc[1], 0, 0, 0);                           \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            if (s == 0) { MID __builtin_amdgcn_sched_barrier(0); }                                             \
            bq = bqn; ak0 = an0; ak1 = an1;                                                                    \
        } }
    VF_A3_LOAD(ka, qa, 0);
    VF_A3_STORE(ka, qa, 0);
    VF_A3_LOAD(ka, qa, CKA);                             // (C >= 32: the dispatcher's C % 32 == 0)
    VF_A3_LOAD(kb, qb_, min(2 * CKA, C - CKA));
    __syncthreads();
    for (int c0 = 0; c0 < C; c0 += 2 * CKA) {            // chunks c0 (buffer 0) and c0 + 16 (buffer 1)
        // behind the first MFMAs: chunk c0 + 16 -> buffer 1 (last read before the barrier), request chunk c0 + 48
        // (clamped at the end: re-reads the last one, unused)
        VF_A3_MFMAS(0, VF_A3_STORE(ka, qa, 1) VF_A3_LOAD(ka, qa, min(c0 + 3 * CKA, C - CKA)));
        VF_LDS_BARRIER();
        VF_A3_MFMAS(1, VF_A3_STORE(kb, qb_, 0) VF_A3_LOAD(kb, qb_, min(c0 + 4 * CKA, C - CKA)));
        VF_LDS_BARRIER();
    }
#undef VF_A3_MFMAS
#undef VF_A3_LOAD
#undef VF_A3_STORE
#undef VF_A3_LK
#undef VF_A3_SK

    // V pieces of this wave: piece n = (channel tile n / 2, key tile n & 1) = 32 rows x 128 bytes, four float4 per lane
    // (8 rows per instruction); piece 0 is requested now and lands during the softmax
#ifdef VF_ATTN_STAMPS
    st_[1] = clock64();
#endif
    f32x4 va0, va1, va2, va3, vb0, vb1, vb2, vb3;        // even pieces | odd pieces: requested two pieces ahead
    // (lane group l / 8 takes rows 4 (l / 8) + I: the four loads of a piece differ by an immediate offset)
    unsigned vvo = (unsigned)((4 * (lane >> 3) * L + w * (NKT * 32) + 4 * (lane & 7)) * 4);    // bytes
    float* const Vw = lds + w * VW;
    float* const vdst = Vw + 4 * (lane >> 3) * RSV + 4 * (lane & 7);
#define VF_A3_LV(R, I, VC) R##I = VF_GLD4(VC, vvo, (I) * (L * 4));
#define VF_A3_LOADV(R, N) { const char* vc_ = uniform_ptr(vp + (size_t)(((N) >> 1) * 32) * L + ((N) & 1) * 32);      \
        asm("" : "+s"(vc_), "+v"(vvo));                                                                          \
        VF_A3_LV(R, 0, vc_) VF_A3_LV(R, 1, vc_) VF_A3_LV(R, 2, vc_) VF_A3_LV(R, 3, vc_) }
#define VF_A3_SV(R, I) *reinterpret_cast<f32x4*>(vdst + (I) * RSV) = R##I;
#define VF_A3_STOREV(R) { VF_A3_SV(R, 0) VF_A3_SV(R, 1) VF_A3_SV(R, 2) VF_A3_SV(R, 3) }
    const int npiece = C / 16;                           // two per 32-channel tile
    VF_A3_LOADV(va, 0);
    VF_A3_LOADV(vb, 1);

    const int qi = q0 + li;
    if constexpr (DSCORE) {
        // acc[t][r] = dP[query qi][key 64 w + 32 t + 8 (r / 4) + 4 lh + r % 4]
        const size_t prow = ((size_t)b * L + qi) * L + w * (NKT * 32) + 4 * lh;
        float4 pv[NKT][4];
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) pv[t][g] = *reinterpret_cast<const float4*>(P + prow + t * 32 + 8 * g);
        float dot = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                dot += (pv[t][g].x * acc[t][4 * g] + pv[t][g].y * acc[t][4 * g + 1]) +
                       (pv[t][g].z * acc[t][4 * g + 2] + pv[t][g].w * acc[t][4 * g + 3]);
        dot += __shfl_xor(dot, 32, 64);
        if (lh == 0) red[0][w][li] = dot;
        __syncthreads();
        VF_A3_STOREV(va);                                // (the staging buffers of the score phase are dead since this barrier)
        VF_A3_LOADV(va, min(2, npiece - 1));
        dot = (red[0][0][li] + red[0][1][li]) + (red[0][2][li] + red[0][3][li]);
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 d = make_float4(pv[t][g].x * (acc[t][4 * g] - dot), pv[t][g].y * (acc[t][4 * g + 1] - dot),
                                             pv[t][g].z * (acc[t][4 * g + 2] - dot), pv[t][g].w * (acc[t][4 * g + 3] - dot));
                *reinterpret_cast<float4*>(dS + prow + t * 32 + 8 * g) = d;
                acc[t][4 * g] = d.x; acc[t][4 * g + 1] = d.y; acc[t][4 * g + 2] = d.z; acc[t][4 * g + 3] = d.w;
            }
    } else {
    // softmax over all keys of the lane's query: this wave's 64 keys, then the other waves' statistics
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[t][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (lh == 0) red[0][w][li] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0][0][li], red[0][1][li]), fmaxf(red[0][2][li], red[0][3][li]));
    const float c2 = alpha * 1.44269504088896341f, mc = mx * c2;     // exp(alpha (s - max)) = 2^(c s - c max), one v_exp_f32
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(acc[t][r], c2, -mc));
            acc[t][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    if (lh == 0) red[1][w][li] = sum;
    VF_A3_STOREV(va);                                    // (the K / Q buffers are dead since the loop's last barrier)
    VF_A3_LOADV(va, min(2, npiece - 1));
    __syncthreads();
    const float inv = 1.0f / ((red[1][0][li] + red[1][1][li]) + (red[1][2][li] + red[1][3][li]));
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] *= inv;

#ifdef VF_ATTN_STAMPS
    st_[2] = clock64();
    if (false) {
#else
    if (P) {
#endif
        float* pr = P + ((size_t)b * L + qi) * L + w * (NKT * 32);
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4*>(pr + t * 32 + 8 * g + 4 * lh) =
                    make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
    }
    }

#ifdef VF_ATTN_STAMPS
    st_[3] = clock64();
#endif
    // O = V P^T.  The wave's LDS instructions execute in order: piece n + 1 is written to the wave's slot behind the last
    // read of piece n, no barrier.  After the two pieces of a channel tile the four partial tiles meet in oex.
    float* const oex = lds + 4 * VW;
    const float* vq = Vw + li * RSV + 4 * lh;
#define VF_A3_PV(T) {                                                                                          \
        float4 av = *reinterpret_cast<const float4*>(vq);                                                      \
        _Pragma("unroll")                                                                                      \
        for (int g = 0; g < 4; ++g) {                                                                          \
            float4 avn = av;                                                                                   \
            if (g + 1 < 4) avn = *reinterpret_cast<const float4*>(vq + (g + 1) * 8);                           \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, acc[T][4 * g + 0], o, 0, 0, 0);                      \
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, acc[T][4 * g + 1], o, 0, 0, 0);                      \
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, acc[T][4 * g + 2], o, 0, 0, 0);                      \
            o = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, acc[T][4 * g + 3], o, 0, 0, 0);                      \
            __builtin_amdgcn_sched_barrier(0);                                                                 \
            av = avn;                                                                                          \
        } }
    for (int ct = 0; ct < npiece / 2; ++ct) {            // channel tile ct = pieces 2 ct (key tile 0), 2 ct + 1 (key tile 1)
        f32x16 o = {0};
        VF_A3_PV(0)
        VF_A3_STOREV(vb);                                // piece 2 ct + 1
        VF_A3_LOADV(vb, min(2 * ct + 3, npiece - 1));    // (clamped at the end: re-read, unused)
        VF_A3_PV(1)
        VF_A3_STOREV(va);                                // piece 2 ct + 2 (at the end: rewritten, unused)
        VF_A3_LOADV(va, min(2 * ct + 4, npiece - 1));
        float* ox = oex + (ct & 1) * OEX;
#pragma unroll
        for (int r = 0; r < 16; ++r) ox[(w * 16 + r) * 64 + lane] = o[r];
        VF_LDS_BARRIER();
        // rows 4 w .. 4 w + 3 of the tile: channels 32 ct + 8 w + (0..3) + 4 lh
        float* ob = out + ((size_t)b * (DSCORE ? 3 * C : C) + ct * 32 + 8 * w + 4 * lh) * L + qi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* x_ = ox + (4 * w + r) * 64 + lane;
            const float v_ = (x_[0] + x_[16 * 64]) + (x_[32 * 64] + x_[48 * 64]);
            ob[(size_t)r * L] = DSCORE ? alpha * v_ : v_;
        }
    }
#undef VF_A3_PV
#undef VF_A3_LOADV
#undef VF_A3_STOREV
#undef VF_A3_LV
#undef VF_A3_SV
#undef VF_GLD4
#ifdef VF_ATTN_STAMPS
    if (tid == 0 && P) {
        long long* o_ = reinterpret_cast<long long*>(P) + (size_t)blockIdx.x * 8;
        o_[0] = st_[0]; o_[1] = st_[1]; o_[2] = st_[2]; o_[3] = st_[3]; o_[4] = clock64(); o_[5] = 0;
        o_[6] = rt0_; o_[7] = wall_clock64();
    }
#endif
}

// Few views: one workgroup = one view x 32 queries, wave w owns keys [32w, 32w+32) -- L/32 waves.  The dependent
// MFMA chain per wave is L/32 times shorter than with one wave per query block and the grid is S x L/32 workgroups:
// the sampler (S = 1..12 views) no longer runs its seven attention layers on a dozen waves (57 -> 21 us per layer).
// With many views this kernel would re-stream K and V once per 32 queries (L2-bound: 113 us at S = 96).
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
    const float c2 = alpha * 1.44269504088896341f, mc = mx * c2;     // exp(alpha (s - max)) = 2^(c s - c max), one v_exp_f32
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[r] = __builtin_amdgcn_exp2f(fmaf(acc[r], c2, -mc));
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

// Very few views (the sampler at B = 1: S <= 16 stacked views), L = 256 (round 5).  In-kernel clocks of the 32-query kernel
// above at S = 1 (tools/attn_split_stamps.py) show what its 22 us are: eight workgroups = eight compute units do ALL the
// work, and each of their SIMDs has 2 x (96 + 96) MFMAs of 64 cycles to issue -- 24.6 k cycles = 10 us of pure matrix
// issue, whatever the staging scheme (LDS chunks, a three-deep register ring, operands straight from global memory, all
// eight workgroups of a view on one XCD: the score phase took 19-21 k cycles every time).  The only way down is MORE
// compute units per view: 16 queries per workgroup -> 16 workgroups per view, on v_mfma_f32_16x16x4_f32:
//   D[16 x 16] += A[16 x 4] B[4 x 16];  lane l: j = l % 16, kk = l / 16;  A[row j][k kk], B[k kk][col j], D[row 4 kk + r][col j]
//   1. S^T tile: A = K^T (rows = 16 keys, k = 4 channels), B = Q (cols = 16 queries); wave w owns keys [32 w, 32 w + 32) =
//      two tiles; both operands are 4-byte loads STRAIGHT from global memory / L2 (for a fixed step the 16 lanes of a
//      quarter read 64 consecutive bytes of one channel row; a wave's K columns are its own), one block of 16 steps ahead
//      of the MFMAs -- no LDS, no barrier in this phase;
//   2. softmax over the 256 keys of the lane's query: 8 registers, two cross-quarter shuffles, then across the waves
//      through LDS (fixed order);
//   3. P -> LDS as [key / 4][query][key % 4]: the lane's four accumulator rows are four consecutive keys = ONE
//      ds_write_b128, and the same float4 is the B operand of four consecutive MFMAs of step 4;
//   4. O tile (16 channels x 16 queries) = V P^T over a key half: 2 C / 16 items of 32 MFMAs, item i = w + 8 r of wave w
//      (every SIMD carries the same number); A = one float4 of V per four MFMAs (k order = 16 blk + 4 kk + i); the two
//      halves of a tile are computed by waves 2k, 2k + 1 and summed through LDS in a fixed order.
// C a multiple of 64.  With many views this kernel would stream K and V 16 times per view: it is the sampler's kernel.
// L = 64 (the 8x8 mid block, C = 320): the same kernel with two waves (32 keys each), four workgroups per view, O items =
// whole 16-channel tiles over all 64 keys (no key halves, no exchange).
template <int L>
__global__ __launch_bounds__(L / 32 * 64) void attn_fwd_q16_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                   float* __restrict__ P, int C, float alpha) {
    constexpr int NW = L / 32, QB = 16, PF = 16;
    constexpr int KPI = L < 128 ? L : 128;           // keys per O item
    constexpr int NG = KPI / 16, NH = L / KPI;       // float4 groups per item (8 / 4), key halves per tile (2 / 1)
    __shared__ __attribute__((aligned(16))) float Pl[(L / 4) * QB * 4];      // [key quad][query 16][4]
    __shared__ float red[2][NW][QB];
    __shared__ __attribute__((aligned(16))) float oxs[NW / 2][4][64];        // partial O tiles of the upper key halves

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int j = lane & 15, kk = lane >> 4;
    const int b = blockIdx.y, q0 = blockIdx.x * QB;
    const float* qb = qkv + (size_t)b * 3 * C * L;
    const float* kb = qb + (size_t)C * L;
    const float* vb = qb + (size_t)2 * C * L;

    // ---- 1. scores ----
    const float* kcol = kb + (size_t)kk * L + 32 * wid + j;          // step s: channel 4 s + kk
    const float* qcol = qb + (size_t)kk * L + q0 + j;
    const int nblk = C / (4 * PF);
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    float kA[PF], kB[PF], qA[PF], lA[PF], lB[PF], rA[PF];            // two register sets: (k tile 0, k tile 1, q) x 2
#define VF_Q16_LOAD(K0, K1, QD, BLK)                                                                          \
    { const size_t o_ = (size_t)min((BLK), nblk - 1) * (4 * PF) * L;     /* (past the end: re-read, unused) */   \
      _Pragma("unroll") for (int s = 0; s < PF; ++s) {                                                         \
          K0[s] = kcol[o_ + (size_t)(4 * s) * L]; K1[s] = kcol[o_ + (size_t)(4 * s) * L + 16];                  \
          QD[s] = qcol[o_ + (size_t)(4 * s) * L]; } }
#define VF_Q16_MMA(K0, K1, QD)                                                                                \
    { _Pragma("unroll") for (int s = 0; s < PF; ++s) {                                                         \
          s0 = __builtin_amdgcn_mfma_f32_16x16x4f32(K0[s], QD[s], s0, 0, 0, 0);                                 \
          s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(K1[s], QD[s], s1, 0, 0, 0); } }
    VF_Q16_LOAD(kA, kB, qA, 0);
    for (int blk = 0; blk < nblk; blk += 2) {
        VF_Q16_LOAD(lA, lB, rA, blk + 1);
        VF_Q16_MMA(kA, kB, qA);
        if (blk + 1 < nblk) {
            VF_Q16_LOAD(kA, kB, qA, blk + 2);
            VF_Q16_MMA(lA, lB, rA);
        }
    }
#undef VF_Q16_LOAD
#undef VF_Q16_MMA

    // first V item of this wave: requested now, lands during the softmax.  item i -> channel tile i / NH, key half i % NH
    const int nitem = NH * (C / 16);                                 // a multiple of NW (C a multiple of 64)
    float4 va0, va1, va2, va3, va4, va5, va6, va7, vn0, vn1, vn2, vn3, vn4, vn5, vn6, vn7;
    va4 = va5 = va6 = va7 = vn4 = vn5 = vn6 = vn7 = make_float4(0.f, 0.f, 0.f, 0.f);     // (unused at L = 64)
#define VF_Q16_LV1(V, G, SRC) if constexpr ((G) < NG) V##G = *reinterpret_cast<const float4*>((SRC) + 16 * (G));
#define VF_Q16_LOADV(V, ITEM) { const int it_ = min((ITEM), nitem - 1);                                        \
        const float* src_ = vb + (size_t)((it_ / NH) * 16 + j) * L + KPI * (it_ % NH) + 4 * kk;                 \
        VF_Q16_LV1(V, 0, src_) VF_Q16_LV1(V, 1, src_) VF_Q16_LV1(V, 2, src_) VF_Q16_LV1(V, 3, src_)             \
        VF_Q16_LV1(V, 4, src_) VF_Q16_LV1(V, 5, src_) VF_Q16_LV1(V, 6, src_) VF_Q16_LV1(V, 7, src_) }
    VF_Q16_LOADV(va, wid);

    // ---- 2. softmax over all keys of the lane's query ----
    float mx = fmaxf(fmaxf(fmaxf(s0[0], s0[1]), fmaxf(s0[2], s0[3])), fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (kk == 0) red[0][wid][j] = mx;
    __syncthreads();
    mx = red[0][0][j];
#pragma unroll
    for (int w = 1; w < NW; ++w) mx = fmaxf(mx, red[0][w][j]);
    const float c2 = alpha * 1.44269504088896341f, mc = mx * c2;     // exp(alpha (s - max)) = 2^(c s - c max)
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c2, -mc));
        s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c2, -mc));
        sum += s0[r] + s1[r];
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (kk == 0) red[1][wid][j] = sum;
    __syncthreads();
    sum = red[1][0][j];
#pragma unroll
    for (int w = 1; w < NW; ++w) sum += red[1][w][j];
    const float inv = 1.0f / sum;
    s0 *= inv;
    s1 *= inv;

    // ---- 3. P -> LDS [key / 4][query][key % 4] (+ global): tile t of wave w = keys 32 w + 16 t + 4 kk + r ----
    *reinterpret_cast<f32x4*>(Pl + ((8 * wid + kk) * QB + j) * 4) = s0;
    *reinterpret_cast<f32x4*>(Pl + ((8 * wid + 4 + kk) * QB + j) * 4) = s1;
    if (P) {
        float* pr = P + ((size_t)b * L + q0 + j) * L + 32 * wid + 4 * kk;
        *reinterpret_cast<f32x4*>(pr) = s0;
        *reinterpret_cast<f32x4*>(pr + 16) = s1;
    }
    __syncthreads();

    // ---- 4. O tile = V[tile] P^T over a key half (L = 256) / over all keys (L = 64) ----
    for (int base = 0; base < nitem; base += NW) {
        const int it = base + wid, t = it / NH, h = it % NH;
        VF_Q16_LOADV(vn, it + NW);                                   // the wave's next item (clamped)
        f32x4 pb[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) pb[g] = *reinterpret_cast<const f32x4*>(Pl + (((KPI / 4) * h + 4 * g + kk) * QB + j) * 4);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#define VF_Q16_PV(G) if constexpr ((G) < NG) {                                                                \
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va##G.x, pb[(G) < NG ? (G) : 0][0], o, 0, 0, 0);                 \
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va##G.y, pb[(G) < NG ? (G) : 0][1], o, 0, 0, 0);                 \
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va##G.z, pb[(G) < NG ? (G) : 0][2], o, 0, 0, 0);                 \
        o = __builtin_amdgcn_mfma_f32_16x16x4f32(va##G.w, pb[(G) < NG ? (G) : 0][3], o, 0, 0, 0); }
        VF_Q16_PV(0) VF_Q16_PV(1) VF_Q16_PV(2) VF_Q16_PV(3) VF_Q16_PV(4) VF_Q16_PV(5) VF_Q16_PV(6) VF_Q16_PV(7)
#undef VF_Q16_PV
        va0 = vn0; va1 = vn1; va2 = vn2; va3 = vn3; va4 = vn4; va5 = vn5; va6 = vn6; va7 = vn7;
        float* ob = out + ((size_t)b * C + t * 16 + 4 * kk) * L + q0 + j;
        if constexpr (NH == 2) {
            if (h == 1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) oxs[wid >> 1][r][lane] = o[r];
            }
            __syncthreads();
            if (h == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) ob[(size_t)r * L] = o[r] + oxs[wid >> 1][r][lane];
            }
            if (base + NW < nitem) __syncthreads();                  // the exchange slots are reused by the next round
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) ob[(size_t)r * L] = o[r];
        }
    }
#undef VF_Q16_LOADV
#undef VF_Q16_LV1
}

// Attention backward at L = 256, the two products that contract over the QUERIES (round 5):
//     dV[c][j] = sum_i dO[c][i] P[i][j]          dK[c][j] = alpha sum_i q[c][i] dS[i][j]
// one launch for both: (product, view) pairs x (C / 64) x 2 workgroups of 64 channels x 128 keys; four waves = 2 x 2, a wave
// owns 32 channels x 64 keys = two accumulators that share the A fragment.  Against the general batched kernel (gemm.hip,
// 64 x 64 tiles, one accumulator per wave, runtime strides and bounds: ~50 scalar / vector instructions and eight exec-mask
// branches per 16 MFMAs): no bounds, every load = scalar base + constant per-lane offset, the first operand reads sit right
// behind the barrier and the next chunk's loads behind the first MFMAs.  A chunk = 32 queries: A image [64 c][32 i (+4)]
// (a lane's four i values = one ds_read_b128, MFMA step e pairs i = e with i = 4 + e like gemm.hip), B image [32 i][128 j (+4)].
// One LDS buffer + register prefetch (26 KB: every workgroup of the launch is resident, 4-5 per compute unit).
// Whole groups of eight (product, view) pairs are dealt one pair per XCD.
__global__ __launch_bounds__(256, 5) void attn_bwd_dvdk_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                               const float* __restrict__ P, const float* __restrict__ dS,
                                                               float* __restrict__ dqkv, int C, float alpha, int S) {
    constexpr int L = 256, BKQ = 32, AS = BKQ + 4, BS = 128 + 4;
    __shared__ __attribute__((aligned(16))) float Al[64 * AS];
    __shared__ __attribute__((aligned(16))) float Bl[BKQ * BS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w & 1, wn = w >> 1, li = lane & 31, lh = lane >> 5;
    const int nt = (C / 64) * 2;                         // tiles per (product, view)
    int pair, tile;
    {
        const int i = blockIdx.x, npair = 2 * S, nfull = (npair >> 3) << 3;
        if (i < nfull * nt) { const int j = i >> 3; pair = ((j / nt) << 3) + (i & 7); tile = j % nt; }
        else { const int r = i - nfull * nt; pair = nfull + r / nt; tile = r % nt; }
    }
    const int b = pair >> 1, z = pair & 1;               // z = 0: dV, 1: dK
    const int m0 = (tile >> 1) * 64, n0 = (tile & 1) * 128;
    const float* A = z ? qkv + (size_t)b * 3 * C * L : dO + (size_t)b * C * L;       // [C][L]: q | dO
    const float* B = (z ? dS : P) + (size_t)b * L * L + n0;                            // [i][j]
    float* out = dqkv + ((size_t)b * 3 + (z ? 1 : 2)) * C * L;

    // staging: A = 64 rows x 128 bytes (thread: row tid / 8 (+ 32), 16 bytes), B = 32 rows x 512 bytes (row tid / 32 + 8 i)
    unsigned avo = (unsigned)(((tid >> 3) * L + 4 * (tid & 7)) * 4), bvo = (unsigned)(((tid >> 5) * L + 4 * (tid & 31)) * 4);
    f32x4 ar0, ar1, br0, br1, br2, br3;
#define VF_GLD4(BASE, OFS, IMM) (*(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) char*)(BASE) + (OFS) + (IMM)))
#define VF_KV_LOAD(K0) {                                                                                        \
        const char* a0_ = uniform_ptr(A + (size_t)m0 * L + (K0));                                                \
        const char* a1_ = uniform_ptr(A + (size_t)(m0 + 32) * L + (K0));                                         \
        const char* b0_ = uniform_ptr(B + (size_t)(K0) * L);                                                     \
        const char* b1_ = uniform_ptr(B + (size_t)((K0) + 8) * L);                                               \
        const char* b2_ = uniform_ptr(B + (size_t)((K0) + 16) * L);                                              \
        const char* b3_ = uniform_ptr(B + (size_t)((K0) + 24) * L);                                              \
        asm("" : "+s"(a0_), "+s"(a1_), "+s"(b0_), "+s"(b1_), "+s"(b2_), "+s"(b3_), "+v"(avo), "+v"(bvo));        \
        ar0 = VF_GLD4(a0_, avo, 0); ar1 = VF_GLD4(a1_, avo, 0);                                                  \
        br0 = VF_GLD4(b0_, bvo, 0); br1 = VF_GLD4(b1_, bvo, 0); br2 = VF_GLD4(b2_, bvo, 0); br3 = VF_GLD4(b3_, bvo, 0); }
    float* const ast = Al + (tid >> 3) * AS + 4 * (tid & 7);
    float* const bst = Bl + (tid >> 5) * BS + 4 * (tid & 31);
#define VF_KV_STORE() {                                                                                         \
        *reinterpret_cast<f32x4*>(ast) = ar0; *reinterpret_cast<f32x4*>(ast + 32 * AS) = ar1;                    \
        *reinterpret_cast<f32x4*>(bst) = br0; *reinterpret_cast<f32x4*>(bst + 8 * BS) = br1;                     \
        *reinterpret_cast<f32x4*>(bst + 16 * BS) = br2; *reinterpret_cast<f32x4*>(bst + 24 * BS) = br3; }

    f32x16 acc0 = {0}, acc1 = {0};
    const float* af = Al + (wm * 32 + li) * AS + 4 * lh;                 // + 8 grp: i = 8 grp + 4 lh + (0..3)
    const float* bf = Bl + (4 * lh) * BS + wn * 64 + li;                 // + (8 grp + e) BS (+ 32: second accumulator)
    VF_KV_LOAD(0);
    for (int k0 = 0; k0 < L; k0 += BKQ) {
        VF_LDS_BARRIER();                                // every wave is done reading the previous chunk
        VF_KV_STORE();
        VF_LDS_BARRIER();
        float4 a_cur = *reinterpret_cast<const float4*>(af);
        float b00 = bf[0], b01 = bf[BS], b02 = bf[2 * BS], b03 = bf[3 * BS];
        float b10 = bf[32], b11 = bf[BS + 32], b12 = bf[2 * BS + 32], b13 = bf[3 * BS + 32];
#pragma unroll
        for (int grp = 0; grp < BKQ / 8; ++grp) {
            float4 a_nxt = a_cur;
            float c00 = b00, c01 = b01, c02 = b02, c03 = b03, c10 = b10, c11 = b11, c12 = b12, c13 = b13;
            if (grp + 1 < BKQ / 8) {
                a_nxt = *reinterpret_cast<const float4*>(af + 8 * (grp + 1));
                const float* bn = bf + 8 * (grp + 1) * BS;
                c00 = bn[0]; c01 = bn[BS]; c02 = bn[2 * BS]; c03 = bn[3 * BS];
                c10 = bn[32]; c11 = bn[BS + 32]; c12 = bn[2 * BS + 32]; c13 = bn[3 * BS + 32];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b00, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b10, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b01, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b11, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (grp == 0) { VF_KV_LOAD(min(k0 + BKQ, L - BKQ)); __builtin_amdgcn_sched_barrier(0); }   // (clamped at the end: unused)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b02, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b12, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b03, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b13, acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt;
            b00 = c00; b01 = c01; b02 = c02; b03 = c03; b10 = c10; b11 = c11; b12 = c12; b13 = c13;
        }
    }
#undef VF_KV_LOAD
#undef VF_KV_STORE
#undef VF_GLD4
    const float sc = z ? alpha : 1.0f;
    float* o = out + (size_t)(m0 + wm * 32 + 4 * lh) * L + n0 + wn * 64 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        o[(size_t)c * L] = sc * acc0[r];
        o[(size_t)c * L + 32] = sc * acc1[r];
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
    // L = 256 (round 5; profiles/r05_attention_q32.md has the sweep):
    //   S <= 16 (the sampler)          16-query kernel: 16 S workgroups
    //   otherwise                      32-query kernel (8 S workgroups of four waves, three per compute unit) or the
    //                                  128-query kernel (2 S workgroups of eight waves, one per compute unit), whichever
    //                                  quantises better on 256 compute units: measured ~17.7 us per started group of
    //                                  32 views against ~66 us per started group of 128 views (S = 96: 54 against 65 us,
    //                                  S = 128: 72 against 67 us, S = 160: 90 against 117 us)
    // (VF_ATTN_Q16=0 / VF_ATTN_Q32=0: tuning aids -- the round-4 choice: key-split kernel up to S = 52, 128-query kernel above)
    static const bool q16 = !(getenv("VF_ATTN_Q16") && getenv("VF_ATTN_Q16")[0] == '0');
    static const bool q32 = !(getenv("VF_ATTN_Q32") && getenv("VF_ATTN_Q32")[0] == '0');
    const bool q32_wins = 177 * ((S + 31) / 32) + 30 < 660 * ((S + 127) / 128);
    if (L == 256 && S <= 16 && C % 64 == 0 && q16)
        hipLaunchKernelGGL(attn_fwd_q16_kernel<256>, dim3(16, S), dim3(512), 0, st, qkv, out, P, C, alpha);
    else if (L == 64 && S <= 16 && C % 64 == 0 && q16)
        hipLaunchKernelGGL(attn_fwd_q16_kernel<64>, dim3(4, S), dim3(128), 0, st, qkv, out, P, C, alpha);
    else if (L == 256 && q32 && q32_wins)
        hipLaunchKernelGGL(attn_fwd_q32_kernel<false>, dim3(8 * S), dim3(256), 0, st, qkv, out, P, C, alpha, S, nullptr, nullptr);
    else if (L == 256 && S <= 52 && !q32)
        hipLaunchKernelGGL(attn_fwd_split_kernel<256>, dim3(8, S), dim3(512), 0, st, qkv, out, P, C, alpha);
    else if (L == 256)
        hipLaunchKernelGGL(attn_fwd_kh_kernel, dim3(2, S), dim3(512), 0, st, qkv, out, P, C, alpha);
    else if (L == 64)
        hipLaunchKernelGGL(attn_fwd_split_kernel<64>, dim3(2, S), dim3(128), 0, st, qkv, out, P, C, alpha);
    else
        return (int)hipErrorInvalidValue;
    VF_RETURN_LAST_ERROR();
}

// Attention backward at L = 256 (C a multiple of 32), first launch (attn_fwd_q32_kernel<true>):
//     dS = P o (dP - rowsum(P o dP)),  dP = dO^T V (never written);    dQ = K dS^T / sqrt(C) -> q third of dqkv
// qkv, dqkv [S][3C][L], dO [S][C][L], P, dS [S][L][L] (dS may not alias P: the dV product still reads P; the dK product
// reads dS).
int vf_attention_dscore(const float* qkv, const float* dO, const float* P, float* dS, float* dqkv, int S, int C, int L,
                        void* stream) {
    if (S <= 0) return 0;
    if (L != 256 || C % 32 != 0 || C < 32) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_fwd_q32_kernel<true>, dim3(8 * S), dim3(256), 0, (hipStream_t)stream, qkv, dqkv,
                       const_cast<float*>(P), C, 1.0f / sqrtf((float)C), S, dO, dS);
    VF_RETURN_LAST_ERROR();
}

// Attention backward at L = 256 (C a multiple of 64), the other launch: dV = dO P and dK = q dS / sqrt(C) -> the v and k
// thirds of dqkv (attn_bwd_dvdk_kernel).  dS is what vf_attention_dscore wrote.
int vf_attention_dvdk(const float* qkv, const float* dO, const float* P, const float* dS, float* dqkv, int S, int C, int L,
                      void* stream) {
    if (S <= 0) return 0;
    if (L != 256 || C % 64 != 0 || C < 64) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_bwd_dvdk_kernel, dim3(2 * S * (C / 64) * 2), dim3(256), 0, (hipStream_t)stream, qkv, dO, P, dS,
                       dqkv, C, 1.0f / sqrtf((float)C), S);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
