// Convolutions of the SAMPLER at few stacked views (reference model/view_fusion.py:179-214 driving model/unet.py's
// convs with S = 1 ... a few views): a 64x64x64-channel layer is 0.3 GFLOP, the chip's matrix cores would be done in
// 2 us -- what such a layer costs is the LATENCY of its longest dependent chain and the number of graph nodes it needs.
// The training-size kernels (conv.hip) fill the chip at small S by splitting K over workgroups and summing the
// partial outputs in a second launch; here the K split lives INSIDE the workgroup, so a layer is ONE node:
//
//   workgroup = 16 RS output channels x 16 output pixels of one view, 8 waves, each wave owning an eighth of the
//   input channels; v_mfma_f32_16x16x4_f32, whose four k-rows are the four lane quarters' current products; the
//   weights are read straight from the UNPACKED OIHW parameter (no packed copy to refresh); the eight partial tiles
//   are summed through LDS in wave order (deterministic), then bias + per-view bias + residual.
//
//   3x3 (conv3_small_kernel): a wave stages the zero-haloed patch of its next 8 channels in its own LDS region (one
//   dword per lane and channel, no workgroup barrier: a wave reads only what it wrote); the 72 products of a round are
//   dealt to the lane quarters in runs of four (one 16-byte weight load per lane and 16 products), their input values
//   found through a per-lane table of LDS addresses.
//   1x1 (conv1_small_kernel): the B values are gathered from global memory directly (16 consecutive pixels per
//   quarter), optionally from the decoder's never-materialised concatenation [x | x2].
#include "common.h"
#include <cstdlib>

namespace {

struct SmallArgs {
    const float* x;
    const float* x2;
    const float* w;       // OIHW, unpacked
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    int S, Cin, C1, Cout, logW, n;     // output map W = H = 1 << logW; 1x1: n = products per wave (multiple of 16)
    // 3x3 only: the residual 1x1 convolution of the block folded in as extra K (vf_conv_small_res), or rx = null:
    //   y += rw[co][:] . [rx | rx2][:, pixel] + rbias[co]        (rC channels in all, the first rC1 from rx)
    const float* rx;
    const float* rx2;
    const float* rw;
    const float* rbias;
    int rC, rC1, rn;                   // rn = folded products per wave (multiple of 16)
    // GroupNorm around the conv without a GroupNorm launch (round 5, vf_conv_small_gn):
    //   ost != null: the launch also accumulates per-(view, output channel) sums of y and y^2 -- what the GroupNorm
    //                BEHIND this conv needs -- as 64-bit fixed-point integers (2^-24 units) with integer atomics: integer
    //                addition is associative, so the result does not depend on the order the workgroups arrive in
    //                (bit-reproducible, unlike float atomics);
    //   ist != null: x is the RAW input of a GroupNorm(+Swish) whose statistics the producer of x left in ist: the
    //                staging applies (x - mean_g) * rstd_g * gamma_c + beta_c [, Swish] on the way in.
    unsigned long long* ost;           // [S][Cout][2], zero before the launch
    const unsigned long long* ist;     // [S][Cin][2]
    const float* ig;                   // gamma, beta of the input GroupNorm
    const float* ib;
    int icpg, isilu;                   // channels per group
    float iinv, ieps;                  // 1 / (2^24 * icpg * H * W)
    int up;                            // 3x3 only: x is stored at half size, nearest-x2-upsampled on read (Upsample conv, unet.py:185-190)
    const float* wp;                   // 3x3 only: the weights in the packed order of vf_conv_small_pack, or null (unpacked OIHW)
};

constexpr float SM_FIX = 16777216.f;   // 2^24: fixed-point unit of the statistics

// mean, rstd * gamma, beta of channel c of view s from the producer's integer sums (biased variance, eps inside the
// square root, like nn.GroupNorm; E[x^2] - mean^2 in fp32 on exactly accumulated sums)
__device__ __forceinline__ void small_gn_coef(const SmallArgs& a, int s, int c, float& mean, float& scale, float& beta) {
    const int g0 = (c / a.icpg) * a.icpg;
    const unsigned long long* p = a.ist + ((size_t)s * a.Cin + g0) * 2;
    long long s1 = 0, s2 = 0;
    for (int i = 0; i < a.icpg; ++i) { s1 += (long long)p[2 * i]; s2 += (long long)p[2 * i + 1]; }
    mean = (float)((double)s1 * (double)a.iinv);
    const float ex2 = (float)((double)s2 * (double)a.iinv);
    const float var = fmaxf(ex2 - mean * mean, 0.f);
    scale = a.ig[c] / sqrtf(var + a.ieps);
    beta = a.ib[c];
}
__device__ __forceinline__ float small_gn_apply(float x, float mean, float scale, float beta, int silu) {
    const float t = fmaf(x - mean, scale, beta);
    return silu ? silu_f(t) : t;
}

constexpr int SM_PX = 16;

// Epilogue operands (bias + per-view bias + residual of the output element this thread will finish) are requested at
// kernel start: their latency runs under the main loop instead of behind the workgroup's last barrier.
// Round 5: every load here is UNCONDITIONAL (clamped indices, a valid stand-in address for an absent operand, the value
// dropped by a select) and the three values are only summed in the epilogue.  Before, the predicated loads sat in
// branches and `add` was formed at once: the compiler put an s_waitcnt vmcnt(0) behind them, and the same for the
// predicated patch / weight loads of the rounds -- four to five global round trips in sequence (~1 us each at these
// occupancies) in front of the first MFMA of a 7 us kernel.
template <int RS>
struct SmallEpi {
    float b0, b1, r;  // bias, per-view bias, residual (or the folded residual conv's bias)
    size_t o;         // output index, or ~0 for a thread without an output element
    __device__ __forceinline__ void fetch(const SmallArgs& a, int s, int cot, int pt) {
        constexpr int TCO = 16 * RS;
        const int tid = threadIdx.x, HW = 1 << (2 * a.logW);
        const int co = cot * TCO + (tid >> 4), op = pt * SM_PX + (tid & 15);
        const int coc = min(co, a.Cout - 1);
        const size_t oc = ((size_t)s * a.Cout + coc) * HW + op;                 // a valid element for every thread
        o = (tid < TCO * SM_PX && co < a.Cout) ? oc : ~(size_t)0;
        // (raw values: the selects that drop a stand-in are part of add(), i.e. of the epilogue -- a select here would be
        //  the first USE of the loads and put their wait in front of the main loop)
        b0 = *(a.bias ? a.bias + coc : a.w);
        b1 = *(a.vbias ? a.vbias + (size_t)s * a.Cout + coc : a.w);
        r = *(a.res ? a.res + oc : (a.rbias ? a.rbias + coc : a.w));
    }
    __device__ __forceinline__ float add(const SmallArgs& a) const {
        return ((a.bias ? b0 : 0.f) + (a.vbias ? b1 : 0.f)) + ((a.res || a.rbias) ? r : 0.f);
    }
};

// the eight waves' partial tiles -> LDS [wave][co 16 RS][px 16] -> fixed-order sum + epilogue
template <int RS>
__device__ __forceinline__ void small_epilogue(const SmallArgs& a, float* red, const f32x4* acc, const SmallEpi<RS>& ep) {
    constexpr int TCO = 16 * RS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kk = lane >> 4;
    float* rw = red + w * (TCO * SM_PX);
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
#pragma unroll
        for (int r = 0; r < 4; ++r) rw[(16 * rs + 4 * kk + r) * SM_PX + j] = acc[rs][r];      // lane holds D[4 kk + r][j]
    __syncthreads();
    float v = 0.f;
    if (ep.o != ~(size_t)0) {
        v = red[tid];
#pragma unroll
        for (int i = 1; i < 8; ++i) v += red[i * (TCO * SM_PX) + tid];
        v += ep.add(a);
        a.y[ep.o] = v;
    }
    if (a.ost && tid < TCO * SM_PX) {              // (whole waves: TCO * 16 is a multiple of 64)
        // sums of this tile's 16 pixels per channel (16 consecutive lanes), then two integer atomics per channel
        float s1 = v, s2 = v * v;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
        if ((tid & 15) == 0 && ep.o != ~(size_t)0) {
            unsigned long long* d = a.ost + (ep.o >> (2 * a.logW)) * 2;      // ep.o / HW = s * Cout + co
            atomicAdd(d, (unsigned long long)__float2ll_rn(s1 * SM_FIX));
            atomicAdd(d + 1, (unsigned long long)__float2ll_rn(s2 * SM_FIX));
        }
    }
}

// ---- 1x1: wave w owns products (= input channels) [w n, (w+1) n), n a multiple of 16; in its q-th group of 16 the lane
// quarter kk holds the channels 16 q + 4 kk + (0..3): one 16-byte weight load per lane, the four quarters of a weight
// row 64 contiguous bytes; B = the lane's pixel of those four channel planes
constexpr int SM_NB = 4;
constexpr int SM_GN_MAXC = 1024;       // most input channels of a 1x1 conv with the GroupNorm applied on load
template <bool CAT>
__global__ __launch_bounds__(512, 4) void conv1_small_kernel(SmallArgs a) {       // (<= 128 VGPRs: two workgroups per CU)
    __shared__ float red[8 * 32 * SM_PX];
    __shared__ float gtab[3 * SM_GN_MAXC];         // input GroupNorm: mean | rstd gamma | beta per channel
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kk = lane >> 4;
    const int HW = 1 << (2 * a.logW);
    const int ptiles = HW / SM_PX, ncot = (a.Cout + 31) / 32;
    int b = blockIdx.x;
    const int pt = b % ptiles; b /= ptiles;
    const int cot = b % ncot;
    const int s = b / ncot;
    SmallEpi<2> ep;
    ep.fetch(a, s, cot, pt);
    const int K = a.Cin;
    const bool gn = !CAT && a.ist != nullptr;      // (kernel-uniform)
    if (gn) {
        for (int c = tid; c < K; c += 512) small_gn_coef(a, s, c, gtab[c], gtab[SM_GN_MAXC + c], gtab[2 * SM_GN_MAXC + c]);
        __syncthreads();
    }
    const int kw = w * a.n + 4 * kk;
    // A rows of this lane (rows past Cout read the last row; their results are never stored)
    const float* wa0 = a.w + (size_t)min(cot * 32 + j, a.Cout - 1) * K;
    const float* wa1 = a.w + (size_t)min(cot * 32 + 16 + j, a.Cout - 1) * K;
    const float* xs = a.x + (size_t)s * (CAT ? a.C1 : a.Cin) * HW + pt * SM_PX + j;
    const float* xs2 = CAT ? a.x2 + (size_t)s * (a.Cin - a.C1) * HW + pt * SM_PX + j : nullptr;

    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int mb = 0; mb < a.n; mb += 16 * SM_NB) {
        f32x4 A0[SM_NB], A1[SM_NB];
        float B[4 * SM_NB];
#pragma unroll
        for (int q = 0; q < SM_NB; ++q) {
            const int k = kw + mb + 16 * q;
            const bool ok = mb + 16 * q < a.n && k < K;
            const int kc = min(k, K - 4);             // (unconditional loads: clamped index, value dropped by a select)
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wa0 + kc), w1 = *reinterpret_cast<const f32x4*>(wa1 + kc);
            A0[q] = ok ? w0 : (f32x4){0.f, 0.f, 0.f, 0.f};
            A1[q] = ok ? w1 : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ci = kc + e;
                const float* src = (CAT && ci >= a.C1) ? xs2 + (size_t)(ci - a.C1) * HW : xs + (size_t)ci * HW;
                const float v = *src;
                B[4 * q + e] = ok ? v : 0.f;
            }
        }
        if (gn) {
#pragma unroll
            for (int q = 0; q < SM_NB; ++q) {
                const int k = kw + mb + 16 * q;
                if (mb + 16 * q < a.n && k < K) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        B[4 * q + e] = small_gn_apply(B[4 * q + e], gtab[k + e], gtab[SM_GN_MAXC + k + e],
                                                      gtab[2 * SM_GN_MAXC + k + e], a.isilu);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < SM_NB; ++q) {
            if (mb + 16 * q < a.n) {              // (wave-uniform)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[q][e], B[4 * q + e], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[q][e], B[4 * q + e], acc[1], 0, 0, 0);
                }
            }
        }
    }
    small_epilogue<2>(a, red, acc, ep);
}

// ---- 3x3, stride 1.  LOGTC: log2 of the tile's columns (4: maps >= 16 wide, one row segment; 3: 8x8 maps, two rows;
// 2: 4x4 maps, the whole map).  Cin a multiple of 32: every wave owns Cin/8 channels (a multiple of 4).
// A round = 8 channels of the wave = 72 products, consecutive in the OIHW weight row.  In the q-th group of 16 products
// (q < 5; the last group is half empty) the lane quarter kk holds products 16 q + 4 kk + (0..3): ONE 16-byte weight load
// per lane and group, the four quarters of a row 64 contiguous bytes -- the scattered 4-byte loads of a channel-per-
// quarter assignment cost 3-4x the address-coalescing cycles, and the texture path is what a workgroup of this kernel
// waits for.  The matching B values come from the wave's LDS patch through a per-lane table of 20 LDS addresses
// (channel and tap of product 16 q + 4 kk + e), computed once per workgroup.
// The loads are NOT double buffered: two register sets cost the second resident workgroup, which hides more latency
// than the prefetch did (measured).
// PACKED: the weights come from the copy made by vf_conv_small_pack -- [16-row block][wave 8][round][group 5][lane 64]
// float4, i.e. every weight load of a wave is 1 KB of consecutive memory (8 cache lines) instead of sixteen 64-byte pieces
// of sixteen OIHW rows.  Round 5: the rounds of this kernel wait for the texture path, not for latency -- with every lane
// of a quarter reading the SAME row (timing experiment) a round costs 4.0 instead of 6.0 us at 64x64, 2.3 instead of 3.1 at
// 32x32 (profiles/r05_sampler.md).  The sampler's weights are static during a generate() call: packed once per weight version.
template <int LOGTC, int RS, bool PACKED>
__global__ __launch_bounds__(512) void conv3_small_kernel(SmallArgs a) {
    constexpr int TC = 1 << LOGTC, TR = SM_PX / TC, PC = TC + 2, PR = TR + 2, PS = PR * PC;     // 54 / 40 / 36
    constexpr int RND = 8, NQ = 5;                 // channels per wave and round; groups of 16 products (72 -> 80)
    constexpr int TCO = 16 * RS;
    constexpr int PATCH = 8 * RND * PS, REDF = 8 * TCO * SM_PX;
    __shared__ float lds[PATCH > REDF ? PATCH : REDF];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, j = lane & 15, kk = lane >> 4;
    const int W = 1 << a.logW, HW = W * W;
    const int ptiles = HW / SM_PX, ncot = (a.Cout + TCO - 1) / TCO;
    int b = blockIdx.x;
    const int pt = b % ptiles; b /= ptiles;
    const int cot = b % ncot;
    const int s = b / ncot;
    SmallEpi<RS> ep;          // (its three loads are issued in round 0 BEHIND the round's own loads: the compiler guards the
                              //  loop's address registers with an s_waitcnt vmcnt(0) at the top of every round, which
                              //  would otherwise make round 0 wait for them before it has requested anything)
    const int K = a.Cin * 9;
    const int cw = a.Cin >> 3;                     // channels per wave (multiple of 4)
    const int c0 = w * cw;
    // tile origin
    const int p0 = pt * SM_PX, ty0 = p0 >> a.logW, tx0 = p0 & (W - 1);
    // staging duty: lane l < PS owns patch element (pr, pc) of every channel
    const int pr = lane / PC, pc = lane - pr * PC;
    const int gy = ty0 + pr - 1, gx = tx0 + pc - 1;
    const bool pin = lane < PS && (unsigned)gy < (unsigned)W && (unsigned)gx < (unsigned)W;
    // (Upsample conv: the padding is padding of the UPSAMPLED map -- bounds in output coordinates, source pixel (gy/2, gx/2))
    const int HWi = a.up ? HW >> 2 : HW;
    const float* xp = a.x + ((size_t)s * a.Cin + c0) * HWi + (pin ? (a.up ? (gy >> 1) * (W >> 1) + (gx >> 1) : gy * W + gx) : 0);
    float* pl = lds + w * (RND * PS);
    // B table: LDS address of product 16 q + 4 kk + e of a round, for this lane's pixel
    typedef const __attribute__((address_space(3))) float* lds_cptr;      // (32-bit: a generic pointer takes two VGPRs)
    lds_cptr tb[NQ][4];
    {
        lds_cptr bl = (lds_cptr)(pl + (j >> LOGTC) * PC + (j & (TC - 1)));
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = min(16 * q + 4 * kk + e, 9 * RND - 1);      // (slots past the round: any valid address)
                const int ci = (k * 57) >> 9;                  // k / 9 for k < 144
                const int tap = k - 9 * ci, dy = (tap * 11) >> 5;          // tap / 3
                tb[q][e] = bl + ci * PS + dy * PC + (tap - 3 * dy);
            }
    }
    // A: this lane's 16-byte piece of each product group
    const float* wa[RS];
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
        wa[rs] = a.w + (size_t)min(cot * TCO + 16 * rs + j, a.Cout - 1) * K + (size_t)c0 * 9 + 4 * kk;

    const int nrp = (cw + RND - 1) / RND;
    const f32x4* wpk[RS];
#pragma unroll
    for (int rs = 0; rs < RS; ++rs)
        wpk[rs] = reinterpret_cast<const f32x4*>(a.wp) + (((size_t)(cot * RS + rs) * 8 + w) * nrp) * (NQ * 64) + lane;

    f32x4 acc[RS];
#pragma unroll
    for (int rs = 0; rs < RS; ++rs) acc[rs] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // input GroupNorm applied on load: lane l < cw works out the coefficients of the wave's channel c0 + l once
    // (cw = Cin / 8 <= 64 channels per wave); the staging below fetches them with v_readlane (the channel is wave-uniform)
    const bool gn = a.ist != nullptr;              // (kernel-uniform)
    float gmean = 0.f, gscale = 1.f, gbeta = 0.f;
    if (gn && lane < cw) small_gn_coef(a, s, c0 + lane, gmean, gscale, gbeta);
    const int nr = (cw + RND - 1) / RND;
    // Rounds (macros on named arrays: everything stays in registers).  Measured per additional round (tools/small_conv_rounds.py,
    // S = 1, chain replay): 6.0 us at 64x64, 3.2 at 32x32, 1.8 at 16x16 -- against 2.4 / 1.2 / 0.6 us of MFMA issue.  It is NOT
    // the load latency: requesting round r + 1 before the MFMAs of round r (a second register set) made every shape 5-20 %
    // SLOWER (profiles/r05_sampler.md); what a round pays for is the texture path -- every 16-byte weight load touches 16
    // rows of the OIHW tensor -- and the LDS hand-over.  Reading the round's B values from LDS in two batches ahead of the
    // MFMAs (instead of four at a time between them) is worth 3-7 %.
#define VF_S3_LOAD(P_, A_, R_)                                                                              \
    {                                                                                                       \
        const int np_ = 9 * min(RND, cw - (R_) * RND);                                                      \
        _Pragma("unroll") for (int i = 0; i < RND; ++i)           /* (unconditional: xp points at a valid element) */ \
            P_[i] = xp[(size_t)min((R_) * RND + i, cw - 1) * HWi];                                           \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
            /* (unconditional: a group past the round's products re-reads the round's first group) */       \
            const size_t wo = (size_t)(R_) * (RND * 9) + (16 * q + 4 * kk < np_ ? 16 * q : 0);              \
            _Pragma("unroll") for (int rs = 0; rs < RS; ++rs)                                               \
                A_[rs][q] = PACKED ? wpk[rs][((R_) * NQ + q) * 64] : *reinterpret_cast<const f32x4*>(wa[rs] + wo); \
        }                                                                                                   \
    }
#define VF_S3_COMPUTE(P_, A_, R_)                                                                           \
    {                                                                                                       \
        const int np = 9 * min(RND, cw - (R_) * RND);      /* products of this round (72 or 36) */          \
        if (gn) {                                  /* (the zero padding is padding of the NORMALISED map: stays zero) */ \
            _Pragma("unroll") for (int i = 0; i < RND; ++i) {                                               \
                const int c = min((R_) * RND + i, cw - 1);                                                  \
                const float m_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gmean), c)); \
                const float s_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gscale), c)); \
                const float b_ = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gbeta), c)); \
                P_[i] = (pin && (R_) * RND + i < cw) ? small_gn_apply(P_[i], m_, s_, b_, a.isilu) : 0.f;    \
            }                                                                                               \
        } else {                                                                                            \
            _Pragma("unroll") for (int i = 0; i < RND; ++i) P_[i] = (pin && (R_) * RND + i < cw) ? P_[i] : 0.f; \
        }                                                                                                   \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
            const bool ok = 16 * q + 4 * kk < np;                                                           \
            _Pragma("unroll") for (int rs = 0; rs < RS; ++rs) A_[rs][q] = ok ? A_[rs][q] : (f32x4){0.f, 0.f, 0.f, 0.f}; \
        }                                                                                                   \
        if (lane < PS) {                                                                                    \
            _Pragma("unroll") for (int i = 0; i < RND; ++i) pl[i * PS + lane] = P_[i];   /* (channels past the wave's range: zeros) */ \
        }                                                                                                   \
        /* (a wave reads only what it wrote itself: LDS operations of one wave execute in order, no workgroup barrier) */ \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                                              \
        __builtin_amdgcn_wave_barrier();                                                                    \
        /* B values: groups 0-2 are read from LDS in one batch in front of their MFMAs, groups 3-4 while those run (one    \
           exposed LDS latency per round instead of five; a slot past the round's products reads a valid address and     \
           meets a zero weight) */                                                                         \
        float bv_[NQ][4];                                                                                   \
        _Pragma("unroll") for (int q = 0; q < 3; ++q)                                                       \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv_[q][e] = *tb[q][e];                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        _Pragma("unroll") for (int q = 3; q < NQ; ++q)                                                      \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) bv_[q][e] = *tb[q][e];                            \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
            if (16 * q < np) {                     /* (wave-uniform) */                                      \
                _Pragma("unroll") for (int e = 0; e < 4; ++e)                                               \
                    _Pragma("unroll") for (int rs = 0; rs < RS; ++rs)                                       \
                        acc[rs] = __builtin_amdgcn_mfma_f32_16x16x4f32(A_[rs][q][e], bv_[q][e], acc[rs], 0, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
        __builtin_amdgcn_wave_barrier();           /* the next round's patch overwrites what this round read */ \
    }
    {
        f32x4 A0[RS][NQ];
        float P0[RND];
        for (int r = 0; r < nr; ++r) {
            VF_S3_LOAD(P0, A0, r);
            if (r == 0) ep.fetch(a, s, cot, pt);     // (behind round 0's loads)
            __builtin_amdgcn_sched_barrier(0);       // every load of the round is in flight before the first value is used
            VF_S3_COMPUTE(P0, A0, r);
        }
    }
#undef VF_S3_LOAD
#undef VF_S3_COMPUTE
    // ---- the block's residual 1x1 convolution as extra K (reference unet.py:238,245: block2(...) + res_conv(x)): the
    // wave's share of the rC residual-input channels, products dealt as in conv1_small_kernel (one 16-byte weight load
    // per lane and 16 products, B = the lane's pixel of four channel planes, straight from global memory -- a tile's 16
    // pixels are consecutive in every map geometry this kernel takes).  One graph node less per residual block.
    if (a.rx) {
        constexpr int NBR = 4;
        const float* wr[RS];
#pragma unroll
        for (int rs = 0; rs < RS; ++rs) wr[rs] = a.rw + (size_t)min(cot * TCO + 16 * rs + j, a.Cout - 1) * a.rC;
        const float* xs = a.rx + (size_t)s * a.rC1 * HW + p0 + j;
        const float* xs2 = a.rx2 ? a.rx2 + (size_t)s * (a.rC - a.rC1) * HW + p0 + j : xs;
        const int kw = w * a.rn + 4 * kk;
        for (int mb = 0; mb < a.rn; mb += 16 * NBR) {
            f32x4 A[RS][NBR];
            float B[4 * NBR];
#pragma unroll
            for (int q = 0; q < NBR; ++q) {
                const int k = kw + mb + 16 * q;
                const bool ok = mb + 16 * q < a.rn && k < a.rC;
                const int kc = min(k, a.rC - 4);          // (unconditional loads, as everywhere in this file)
#pragma unroll
                for (int rs = 0; rs < RS; ++rs) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(wr[rs] + kc);
                    A[rs][q] = ok ? wv : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int ci = kc + e;
                    const float v = *(ci >= a.rC1 ? xs2 + (size_t)(ci - a.rC1) * HW : xs + (size_t)ci * HW);
                    B[4 * q + e] = ok ? v : 0.f;
                }
            }
#pragma unroll
            for (int q = 0; q < NBR; ++q) {
                if (mb + 16 * q < a.rn) {          // (wave-uniform)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int rs = 0; rs < RS; ++rs)
                            acc[rs] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[rs][q][e], B[4 * q + e], acc[rs], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();                               // the reduction buffer overlays the patches
    small_epilogue<RS>(a, lds, acc, ep);
}

template <int RS, bool PACKED>
int launch_conv3(const SmallArgs& a, hipStream_t st) {
    const int W = 1 << a.logW;
    const long grid = (long)a.S * ((a.Cout + 16 * RS - 1) / (16 * RS)) * (W * W / SM_PX);
    if (W >= 16) hipLaunchKernelGGL((conv3_small_kernel<4, RS, PACKED>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else if (W == 8) hipLaunchKernelGGL((conv3_small_kernel<3, RS, PACKED>), dim3((unsigned)grid), dim3(512), 0, st, a);
    else hipLaunchKernelGGL((conv3_small_kernel<2, RS, PACKED>), dim3((unsigned)grid), dim3(512), 0, st, a);
    VF_RETURN_LAST_ERROR();
}

// [16-row block 2 ceil(Cout/32)][wave 8][round nr][group 5][lane 64] float4: lane (j, kk) of group q, round r, wave wv holds
// W[16 blk + j][(wv cw + 8 r) 9 + 16 q + 4 kk + (0..3)], zero past the round's products / past Cout
__global__ __launch_bounds__(256) void conv3_small_pack_kernel(const float* __restrict__ w, float4* __restrict__ out, int Cout,
                                                                int Cin, long n4) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n4) return;
    const int cw = Cin >> 3, nr = (cw + 7) / 8;
    const int lane = (int)(idx & 63), q = (int)((idx >> 6) % 5), r = (int)((idx / 320) % nr);
    const int wv = (int)((idx / (320L * nr)) & 7), blk = (int)(idx / (320L * nr * 8));
    const int j = lane & 15, kk = lane >> 4, co = blk * 16 + j;
    const int np = 9 * min(8, cw - r * 8), k0 = 16 * q + 4 * kk;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (co < Cout && k0 < np) {
        const float* p = w + (size_t)co * Cin * 9 + (size_t)(wv * cw + r * 8) * 9 + k0;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = p[e];
    }
    out[idx] = make_float4(v[0], v[1], v[2], v[3]);
}

int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

}  // namespace

extern "C" {

// 1 if vf_conv_small handles this layer: square power-of-two output maps of at least 16 pixels, stride 1 (mode 0) or, 3x3
// only, the nearest-x2-upsampled input of the Upsample conv (mode 2: x stored at H/2 x W/2; round 5); 1x1 with Cin a
// multiple of 4 (16-byte weight loads straight from the OIHW tensor), 3x3 with Cin a multiple of 32 (whole channel
// groups per wave)
int vf_conv_small_supported(int Cin, int Cout, int H, int W, int KS, int mode) {
    if (H != W || H < 4 || (H & (H - 1)) || Cin < 1 || Cout < 1 || (mode != 0 && !(mode == 2 && KS == 3 && H >= 8))) return 0;
    if (KS == 1) return Cin % 4 == 0;
    if (KS == 3) return Cin % 32 == 0;
    return 0;
}

// y = conv(x [| x2 on channels C1..], w) + bias + view_bias + residual at the sampler's sizes (stride 1, H = W).
// w: the unpacked OIHW parameter.  One launch, no workspace.
struct SmallGn {                       // GroupNorm on the input (applied on load) / statistics of the output
    const unsigned long long* ist = nullptr;
    const float* ig = nullptr;
    const float* ib = nullptr;
    int groups = 0, silu = 0;
    float eps = 0.f;
    unsigned long long* ost = nullptr;
    const float* wp = nullptr;         // packed 3x3 weights (vf_conv_small_pack), or null
};

static int conv_small_launch(const float* x, const float* x2, int C1, const float* w, const float* bias,
                             const float* view_bias, const float* residual, float* y, int S, int Cin, int Cout, int H,
                             int W, int KS, int mode, const float* rx, const float* rx2, int rC1, int rC,
                             const float* rw, const float* rbias, void* stream, const SmallGn& gn = SmallGn()) {
    if (!vf_conv_small_supported(Cin, Cout, H, W, KS, mode) || (x2 && (KS != 1 || C1 <= 0 || C1 >= Cin)))
        return (int)hipErrorInvalidValue;
    if (gn.ist && (x2 || !gn.ig || !gn.ib || gn.groups <= 0 || Cin % gn.groups != 0 || (KS == 1 && Cin > SM_GN_MAXC)
                   || (KS == 3 && Cin / 8 > 64)))
        return (int)hipErrorInvalidValue;
    if (mode == 2 && (rx || gn.ist)) return (int)hipErrorInvalidValue;
    if (rx && (KS != 3 || residual || !rw || rC < 4 || rC % 4 != 0 || (rx2 ? (rC1 <= 0 || rC1 >= rC) : rC1 != rC)))
        return (int)hipErrorInvalidValue;
    if (S <= 0) return 0;
    SmallArgs a;
    a.x = x; a.x2 = x2; a.w = w; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.S = S; a.Cin = Cin; a.C1 = x2 ? C1 : Cin; a.Cout = Cout; a.logW = ilog2(W);
    a.n = (((Cin + 7) / 8) + 15) & ~15;
    a.rx = rx; a.rx2 = rx2; a.rw = rw; a.rbias = rbias; a.rC = rC; a.rC1 = rC1;
    a.rn = rx ? ((((rC + 7) / 8) + 15) & ~15) : 0;
    a.ost = gn.ost; a.ist = gn.ist; a.ig = gn.ig; a.ib = gn.ib;
    a.icpg = gn.ist ? Cin / gn.groups : 1; a.isilu = gn.silu; a.ieps = gn.eps;
    a.iinv = gn.ist ? (float)(1.0 / (16777216.0 * (double)a.icpg * (double)H * (double)W)) : 0.f;
    a.up = mode == 2 ? 1 : 0;
    a.wp = KS == 3 ? gn.wp : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (KS == 1) {
        const long grid = (long)S * ((Cout + 31) / 32) * (H * W / SM_PX);
        if (x2) hipLaunchKernelGGL((conv1_small_kernel<true>), dim3((unsigned)grid), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((conv1_small_kernel<false>), dim3((unsigned)grid), dim3(512), 0, st, a);
        VF_RETURN_LAST_ERROR();
    }
    // 32-channel tiles when they alone fill the chip, 16-channel tiles (twice the workgroups) otherwise
    const long wgs32 = (long)S * ((Cout + 31) / 32) * (H * W / SM_PX);
    if (a.wp) return wgs32 >= 256 ? launch_conv3<2, true>(a, st) : launch_conv3<1, true>(a, st);
    return wgs32 >= 256 ? launch_conv3<2, false>(a, st) : launch_conv3<1, false>(a, st);
}

int vf_conv_small(const float* x, const float* x2, int C1, const float* w, const float* bias, const float* view_bias,
                  const float* residual, float* y, int S, int Cin, int Cout, int H, int W, int KS, int mode, void* stream) {
    return conv_small_launch(x, x2, C1, w, bias, view_bias, residual, y, S, Cin, Cout, H, W, KS, mode, nullptr, nullptr, 0,
                             0, nullptr, nullptr, stream);
}

// A residual block's LAST convolution and its residual 1x1 convolution in one launch (3x3, stride 1):
//   y = conv3x3(x, w) + bias + view_bias + conv1x1([rx | rx2], rw) + rbias
// rw: the unpacked (Cout, rC, 1, 1) parameter, rC % 4 == 0; rx2 == NULL: all rC channels from rx (rC1 = rC).
int vf_conv_small_res(const float* x, const float* w, const float* bias, const float* view_bias, float* y, int S, int Cin,
                      int Cout, int H, int W, const float* rx, const float* rx2, int rC1, int rC, const float* rw,
                      const float* rbias, void* stream) {
    if (!rx) return (int)hipErrorInvalidValue;
    return conv_small_launch(x, nullptr, 0, w, bias, view_bias, nullptr, y, S, Cin, Cout, H, W, 3, 0, rx, rx2,
                             rx2 ? rC1 : rC, rC, rw, rbias, stream);
}

// The general form (round 5): the two calls above plus GroupNorm without a GroupNorm launch on either side.
//   in_stats  != NULL: x is the raw input of GroupNorm(in_groups, eps)[+Swish]; in_stats = the [S][Cin][2] integer sums
//                      (y, y^2 in 2^-24 units) that the launch which produced x accumulated; (x - mean) rstd gamma + beta
//                      [, Swish] is applied while x is staged.  Not with x2.
//   out_stats != NULL: [S][Cout][2] 64-bit integers, ZERO before the launch; the launch adds the sums of y and y^2 of
//                      every (view, channel) with integer atomics (order-independent: bit-reproducible).
//   rx != NULL (3x3): the residual 1x1 conv folded in, as vf_conv_small_res.
//   w_packed != NULL (3x3): the weights in the order of vf_conv_small_pack (w is then only the stand-in address).
//   mode: 0, or 2 (3x3: x stored at half size, nearest-x2 upsampled on read; not with in_stats / rx).
int vf_conv_small_gn(const float* x, const float* x2, int C1, const float* w, const float* bias, const float* view_bias,
                     const float* residual, float* y, int S, int Cin, int Cout, int H, int W, int KS,
                     const unsigned long long* in_stats, const float* in_gamma, const float* in_beta, int in_groups,
                     float eps, int silu, unsigned long long* out_stats, const float* rx, const float* rx2, int rC1,
                     int rC, const float* rw, const float* rbias, const float* w_packed, int mode, void* stream) {
    SmallGn gn;
    gn.wp = w_packed;
    gn.ist = in_stats; gn.ig = in_gamma; gn.ib = in_beta; gn.groups = in_groups; gn.silu = silu; gn.eps = eps;
    gn.ost = out_stats;
    return conv_small_launch(x, x2, C1, w, bias, view_bias, residual, y, S, Cin, Cout, H, W, KS, mode, rx, rx2,
                             rx ? (rx2 ? rC1 : rC) : 0, rx ? rC : 0, rw, rbias, stream, gn);
}

// Packed copy of a 3x3 layer's weights for the one-launch kernel (w_packed of vf_conv_small_gn): floats needed / the pack.
long vf_conv_small_pack_floats(int Cout, int Cin) {
    const int cw = Cin >> 3, nr = (cw + 7) / 8;
    return (long)((Cout + 31) / 32) * 2 * 8 * nr * 5 * 64 * 4;      // whole 32-row tiles: a 32-channel workgroup reads two blocks
}
int vf_conv_small_pack(const float* w_oihw, float* w_packed, int Cout, int Cin, void* stream) {
    if (Cin % 32 != 0 || Cout < 1) return (int)hipErrorInvalidValue;
    const long n4 = vf_conv_small_pack_floats(Cout, Cin) / 4;
    hipLaunchKernelGGL(conv3_small_pack_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_oihw,
                       (float4*)w_packed, Cout, Cin, n4);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
