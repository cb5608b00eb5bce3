// Convolution path of the UNet (reference model/unet.py:42,189,198,214,238,255,256) as
// exact-fp32 MFMA implicit GEMMs for gfx950.
//
//   D[co][pixel] = sum_{tap,ci} Wp[tap][ci][co] * X[ci][pixel (+) tap]
//
// * v_mfma_f32_32x32x2_f32: A = packed weights (row = co), B = activations (col = pixel), so
//   the accumulator has the pixel on the lane -> NCHW stores are 128-B coalesced.  fp32 in /
//   fp32 accumulate = a k-ordered fmaf chain, i.e. the reference's arithmetic type.
// * One workgroup (4 waves) = 64 output channels x 128 output pixels (whole image rows, or two
//   8x8 images).  Per K-chunk (8 input channels x all taps; 32 channels for 1x1) the weight
//   slab and the zero-haloed input patch are staged in LDS; the next chunk's global loads are
//   issued into registers before the current chunk's MFMAs (issue-early / write-late).
// * MODE selects how the input patch is addressed, so the same kernel serves
//     0: stride-1 conv / dgrad          1: stride-2 conv (Downsample)
//     2: nearest-x2-upsampled input (Upsample conv, no 4x tensor is materialised)
//     4: sub-pixel dgrad of the stride-2 conv: x = dY (HxW), y = dX (2Hx2W); each output parity (2i+a, 2j+b)
//        has its own accumulator set and tap subset, so no zeros are multiplied
// * Epilogue fuses bias[co] + per-view bias[s][co] (time/angle embedding) + residual.
// * wgrad: D[co][ci] per tap = sum_pixels dY[co][p] * X[ci][p (+) tap]; split-K over pixel
//   tiles into slabs, reduced by a second deterministic kernel (no float atomics).
//
// Bound: fp32 matrix rate (157.3 TF), not HBM: AI of these layers is 144-960 flop/B.
#include "common.h"
#include "wgrad_reduce.h"
#include <cstring>
#include "conv1x1.h"
#include <cstdlib>

namespace {

constexpr int TCO = 64;    // output channels per workgroup tile
constexpr int WGRAD_TARGET_WGS = 512;  // split-K slices are sized for ~2 workgroups per CU

// TPIX = output pixels per workgroup tile (64 / 128 / 256): whole image rows, or several whole
// images when an image has fewer pixels than the tile.
template <int KS, int LOGW, int MODE, int TPIX = 128>
struct Geo {
    static constexpr int W = 1 << LOGW;
    static constexpr int H = W;
    static constexpr int HW = W * H;
    static constexpr int IM = HW >= TPIX ? 1 : TPIX / HW;        // images per tile
    static constexpr int TH = HW >= TPIX ? TPIX / W : H;         // output rows per image per tile
    static constexpr int TPI = HW >= TPIX ? HW / TPIX : 1;       // tiles per image
    static constexpr int PAD = KS / 2;
    static constexpr int PH = MODE == 1 ? 2 * TH + 1 : (MODE == 4 ? TH + 1 : TH + 2 * PAD);
    static constexpr int PW = KS == 1 ? W : (MODE == 1 ? 2 * W + 4 : (MODE == 4 ? W + 4 : W + 8));
    static constexpr int CO = (KS == 1 || MODE == 4) ? 0 : 3;    // LDS column of patch x = -1 (MODE 4: no left halo)
    static constexpr int Q = (MODE == 1 ? 2 * W : W) / 4;        // interior float4 per patch row
    static constexpr int PS = IM * PH * PW;                      // floats per channel plane
    static constexpr int SH = MODE == 1 ? 2 * H : (MODE == 2 ? H / 2 : H);  // source
    static constexpr int SW = MODE == 1 ? 2 * W : (MODE == 2 ? W / 2 : W);
    // LDS offset (within a channel plane) of the patch element that output pixel p (0..127 of
    // the tile) reads for tap (0,0)
    static __host__ __device__ constexpr int pix_off(int p) {
        const int im = IM > 1 ? (p >> (2 * LOGW)) : 0;
        const int q = p & (HW - 1);
        const int r = IM > 1 ? (q >> LOGW) : (p >> LOGW);
        const int c = p & (W - 1);
        return MODE == 1 ? im * PH * PW + 2 * r * PW + 2 * c + CO : im * PH * PW + r * PW + c + CO;
    }
};

struct ConvArgs {
    const float* x;
    const float* w;      // packed [co tile][ci chunk][group][co 64][ci 8]
    const float* bias;   // [Cout] or null
    const float* vbias;  // [S][Cout] or null
    const float* res;    // [S][Cout][H][W] or null
    float* y;
    int S, Cin, Cout, CinP, CoutP;
    int ksplit;   // > 1: split the K (input-channel) loop over ksplit workgroups, partials go to ws
    float* ws;    // [ksplit][S][Cout][H*W] partial sums (small-batch / sampler regime)
    // 1x1 convs on a never-materialised channel concatenation (decoder skip connections): input channels
    // [0, C1) come from x, [C1, Cin) from x2 (x2 == null: plain); output channels [0, C1o) go to y, the rest to y2
    const float* x2;
    float* y2;
    int C1, C1o;
    // inference fusion (vf_conv_fwd_gn): the GroupNorm(+Swish) that consumes this conv's output.  gn_out != null: the
    // split-K reduce kernel also normalises (one launch instead of two); y is then written only if gn_store_y.
    const float* gn_gamma = nullptr;
    const float* gn_beta = nullptr;
    float* gn_out = nullptr;
    float* gn_stats = nullptr;      // [2][S * groups] scratch for the unfused fallback
    int gn_groups = 0, gn_silu = 0, gn_store_y = 1;
    float gn_eps = 1e-5f;
};

// Global -> register load of one float4 of the input patch (zero outside the image / tensor).
// SAFE (1x1 convs whose channel counts fill every 32-channel chunk and whose views fill every tile: the host checks):
// every element of the patch exists, the load is UNCONDITIONAL.  A load behind a (per-lane) condition is a branch
// around it, the compiler then no longer knows how many requests are in flight and its s_waitcnt before the LDS stores
// of the chunk also waits for the youngest ones: 7 % on the 1x1 family (round 4, tools/conv1x1_table.py).
template <class G, int MODE, bool SAFE = false>
__device__ __forceinline__ float4 load_patch4(const float* __restrict__ x, int S, int Cin, int s, int ci,
                                              int r0, int pr, int q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (SAFE && MODE == 0 && G::PAD == 0)
        return *reinterpret_cast<const float4*>(x + ((size_t)s * Cin + ci) * (size_t)(G::SH * G::SW) + (r0 + pr) * G::SW + 4 * q);
    if (s >= S || ci >= Cin) return v;
    const size_t plane = ((size_t)s * Cin + ci) * (size_t)(G::SH * G::SW);
    if (MODE == 0) {
        const int gy = r0 + pr - G::PAD;
        if (gy >= 0 && gy < G::H) v = *reinterpret_cast<const float4*>(x + plane + gy * G::SW + 4 * q);
    } else if (MODE == 1) {
        const int gy = 2 * r0 + pr - 1;
        if (gy >= 0 && gy < G::SH) v = *reinterpret_cast<const float4*>(x + plane + gy * G::SW + 4 * q);
    } else if (MODE == 4) {                       // rows r0 .. r0+TH (one halo row below, zero column on the right)
        const int gy = r0 + pr;
        if (gy < G::H) v = *reinterpret_cast<const float4*>(x + plane + gy * G::SW + 4 * q);
    } else {
        const int uy = r0 + pr - G::PAD;
        if (uy >= 0 && uy < G::H) {
            const float2 t = *reinterpret_cast<const float2*>(x + plane + (uy >> 1) * G::SW + 2 * q);
            v = make_float4(t.x, t.x, t.y, t.y);
        }
    }
    return v;
}

// NCO > 1 (1x1 convs): the workgroup covers NCO consecutive 64-channel weight tiles, so the activation tile
// is staged once for 64*NCO output channels -- a 64-channel workgroup of a 1x1 conv needs ~12 B/clk of
// L2->LDS traffic per CU at full MFMA rate, which is the load path's limit.
template <int KS, int LOGW, int MODE, int NPT, int NCO = 1, bool SAFE = false>
__global__ __launch_bounds__(256, (KS == 1 && NPT == 1 && NCO == 1) ? 5 : 1) void conv_mfma_kernel(ConvArgs a) {
    constexpr int TPIX = 64 * NPT;            // each wave: 32 channels x (32*NPT) pixels
    using G = Geo<KS, LOGW, MODE, TPIX>;
    constexpr int CK = KS == 3 ? 8 : 32;      // input channels per K-chunk
    constexpr int NG = KS == 3 ? 9 : 4;       // 8-channel groups per chunk: 9 taps (3x3) or 4 sub-chunks (1x1)
    constexpr int WROW = 12;                  // LDS floats per (group, co) row: 8 used, stride 12 -> b128 reads conflict-free
    constexpr int NW4 = NG * TCO * 2;         // float4 per weight chunk
    constexpr int NX4 = CK * G::IM * G::PH * G::Q;

    constexpr int WLS = NG * TCO * WROW;      // LDS floats of one weight tile
    __shared__ __attribute__((aligned(16))) float wl[NCO * WLS];
    __shared__ __attribute__((aligned(16))) float xl[CK * G::PS];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, pw = wid >> 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ncot1 = a.CoutP / TCO;                 // 64-channel weight tiles
    const int ncot = (ncot1 + NCO - 1) / NCO;        // workgroup tiles
    const unsigned logical0 = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical0 % a.ksplit;
    const unsigned logical = logical0 / a.ksplit;
    const int cot = logical % ncot;
    const int tile = logical / ncot;
    const int co0 = cot * NCO * TCO;
    const int nco_here = min(NCO, ncot1 - cot * NCO);   // weight tiles this workgroup really has
    const int nch = a.CinP / CK;
    const int cbeg = (split * nch / a.ksplit) * CK, cend = ((split + 1) * nch / a.ksplit) * CK;
    const int s0 = G::IM > 1 ? tile * G::IM : tile / G::TPI;
    const int r0 = G::IM > 1 ? 0 : (tile % G::TPI) * G::TH;

#ifdef VF_CONV_STAMPS   // diagnostic build only (tools/conv_stamps.py): per-workgroup phase clocks
    long long st_[4] = {clock64(), 0, 0, 0}, rt_[2] = {wall_clock64(), 0};
#endif
    // zero the patch once: the left/right halo columns are never written again
    if (KS != 1)                                           // (a 1x1 patch has no halo)
        for (int i = tid; i < CK * G::PS; i += 256) xl[i] = 0.f;

    // Register staging: full 256-wide passes are unconditional and a ragged tail lives in its
    // own scalar so that the arrays are only ever indexed statically (stay in VGPRs).
    constexpr int NWF = NW4 / 256, NXF = NX4 / 256;
    constexpr bool WT = (NW4 % 256) != 0, XT = (NX4 % 256) != 0;
    static_assert(NCO == 1 || !WT, "multi-tile workgroups assume whole 256-wide weight passes");
    // PD register sets = prefetch distance in chunks.  Measured on the 1x1 layers: distance 2 is not faster than 1
    // (tools/conv_stamps.py: the K loop is already MFMA-paced; the exposed phases are the first load and the
    // output store, which every workgroup of a one-round grid reaches at the same time).
    constexpr int PD = 1;
    float4 wreg[PD][NCO * (NWF > 0 ? NWF : 1)], xreg[PD][NXF > 0 ? NXF : 1];
    float4 wtail[PD], xtail[PD];
#pragma unroll
    for (int d = 0; d < PD; ++d) wtail[d] = xtail[d] = make_float4(0.f, 0.f, 0.f, 0.f);
    // packed weights: [co tile][chunk][group][co 64][ci 8] -> one contiguous block per (tile, chunk);
    // a missing trailing tile (Cout not a multiple of 64*NCO) re-reads the last one and is never used
    const size_t wtile = (size_t)(a.CinP / CK) * (NG * TCO * 8);
    const float* wsrc = a.w + (size_t)cot * NCO * wtile;
    auto load_w = [&](int e, int c0, int j = 0) -> float4 {
        return *reinterpret_cast<const float4*>(wsrc + (size_t)min(j, nco_here - 1) * wtile +
                                                (size_t)(c0 / CK) * (NG * TCO * 8) + 4 * e);
    };
    auto load_x = [&](int e, int c0) -> float4 {
        const int q = e % G::Q;
        const int t1 = e / G::Q;
        const int pr = t1 % G::PH;
        const int t2 = t1 / G::PH;
        const int im = t2 % G::IM, ci = t2 / G::IM;
        if (KS == 1 && a.x2) {                                    // concatenated input: chunk-uniform source
            const bool second = c0 >= a.C1;
            return load_patch4<G, MODE, SAFE>(second ? a.x2 : a.x, a.S, second ? a.Cin - a.C1 : a.C1, s0 + im,
                                              second ? c0 - a.C1 + ci : c0 + ci, r0, pr, q);
        }
        return load_patch4<G, MODE, SAFE>(a.x, a.S, a.Cin, s0 + im, c0 + ci, r0, pr, q);
    };
    auto store_x = [&](int e, const float4& v) {
        const int q = e % G::Q;
        const int t1 = e / G::Q;                                  // (ci*IM + im)*PH + pr
        *reinterpret_cast<float4*>(xl + t1 * G::PW + 4 * q + ((KS == 1 || MODE == 4) ? 0 : 4)) = v;
    };
#define VF_LOAD_CHUNK(D_, C0)                                                         \
    {                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NCO; ++j)                                \
            _Pragma("unroll") for (int i = 0; i < NWF; ++i) wreg[D_][j * NWF + i] = load_w(tid + i * 256, (C0), j); \
        if (WT && tid + NWF * 256 < NW4) wtail[D_] = load_w(tid + NWF * 256, (C0));    \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) xreg[D_][i] = load_x(tid + i * 256, (C0)); \
        if (XT && tid + NXF * 256 < NX4) xtail[D_] = load_x(tid + NXF * 256, (C0));    \
    }
#define VF_STORE_CHUNK(D_)                                                            \
    {                                                                                 \
        _Pragma("unroll") for (int j = 0; j < NCO; ++j)                                \
            _Pragma("unroll") for (int i = 0; i < NWF; ++i)                            \
                *reinterpret_cast<float4*>(wl + j * WLS + ((tid + i * 256) >> 1) * WROW + 4 * (tid & 1)) = wreg[D_][j * NWF + i]; \
        if (WT && tid + NWF * 256 < NW4)                                               \
            *reinterpret_cast<float4*>(wl + ((tid + NWF * 256) >> 1) * WROW + 4 * (tid & 1)) = wtail[D_]; \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) store_x(tid + i * 256, xreg[D_][i]); \
        if (XT && tid + NXF * 256 < NX4) store_x(tid + NXF * 256, xtail[D_]);          \
    }

    // MODE 4 (sub-pixel dgrad of the stride-2 conv): four accumulator sets = the four output parities (2i+a, 2j+b);
    // tap (kh, kw) feeds exactly one of them, so the 9 MFMA groups of a chunk are the whole work -- a quarter of
    // what the zero-dilated form (MODE 3) multiplies.
    constexpr int NACC = MODE == 4 ? 4 : NCO;
    static_assert(MODE != 4 || NCO == 1, "MODE 4 uses the accumulator sets for the output parities");
    f32x16 acc[NACC][NPT];
    int xo[NPT];
#pragma unroll
    for (int nt = 0; nt < NPT; ++nt) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) acc[j][nt] = (f32x16){0};
        xo[nt] = 4 * lh * G::PS + G::pix_off(pw * 32 * NPT + nt * 32 + li);
    }
    // k order inside an 8-channel group: MFMA step s pairs channel s (lane half 0) with channel
    // 4+s (half 1), so a lane's four A values are contiguous -> one ds_read_b128 per group.
    const float* wb = wl + (cw * 32 + li) * WROW + 4 * lh;
    auto frag_a = [&](int g, int j = 0) -> float4 {
        return *reinterpret_cast<const float4*>(wb + j * WLS + g * TCO * WROW);
    };
    auto frag_b = [&](int g, int nt, int s) -> float {
        // MODE 4: packed tap g holds w[kh][kw] with kh = 2 - g/3, kw = 2 - g%3 (dgrad pack = flipped taps); it
        // reads dy one row down / one column right iff kh == 0 / kw == 0
        const int off = MODE == 4 ? (g / 3 == 2 ? G::PW : 0) + (g % 3 == 2 ? 1 : 0) + s * G::PS
                                  : (KS == 3 ? (g / 3) * G::PW + (g % 3) + s * G::PS : (8 * g + s) * G::PS);
        return xl[xo[nt] + off];
    };

    // chunk loads run unconditionally on clamped chunk indices (the tail re-reads the last chunk): no branch
    // around a load, so the compiler counts the loads in flight instead of draining the queue
    const int nchk = (cend - cbeg) / CK;
#pragma unroll
    for (int d = 0; d < PD; ++d) VF_LOAD_CHUNK(d, cbeg + min(d, nchk - 1) * CK);
    for (int cb = 0; cb < nchk; cb += PD)
#pragma unroll
    for (int d = 0; d < PD; ++d) {
        if (PD > 1 && cb + d >= nchk) break;
        __syncthreads();                                   // previous chunk's LDS reads (and the zero fill) done
        VF_STORE_CHUNK(d);
        __syncthreads();
#ifdef VF_CONV_STAMPS
        if (cb + d == 0) st_[1] = clock64();
#endif
        VF_LOAD_CHUNK(d, cbeg + min(cb + d + PD, nchk - 1) * CK);
        // software pipeline over the groups: fragments of group g+1 are fetched from LDS before
        // the MFMAs of group g are issued (sched_barrier pins that order)
        float4 a_cur[NCO];
#pragma unroll
        for (int j = 0; j < NCO; ++j) a_cur[j] = frag_a(0, j);
        float b_cur[NPT][4];
#pragma unroll
        for (int nt = 0; nt < NPT; ++nt)
#pragma unroll
            for (int s = 0; s < 4; ++s) b_cur[nt][s] = frag_b(0, nt, s);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            float4 a_nxt[NCO];
            float b_nxt[NPT][4];
#pragma unroll
            for (int j = 0; j < NCO; ++j) a_nxt[j] = a_cur[j];
            // the LDS reads of group g+1 go BEHIND the first MFMA of group g: all waves of a SIMD run this loop in
            // phase, reads in front of the MFMAs would leave the matrix pipe idle in all of them at once
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NCO; ++j) {
                if (NCO > 1 && j >= nco_here) continue;            // wave-uniform: trailing tile absent
                const float av[4] = {a_cur[j].x, a_cur[j].y, a_cur[j].z, a_cur[j].w};
                // output parity of tap g in MODE 4: a = (kh != 1), b = (kw != 1)
                const int ja = MODE == 4 ? 2 * (g / 3 != 1) + (g % 3 != 1) : j;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int nt = 0; nt < NPT; ++nt) {
                        acc[ja][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], b_cur[nt][s], acc[ja][nt], 0, 0, 0);
                        if (j == 0 && s == 0 && nt == 0) {
                            __builtin_amdgcn_sched_barrier(0);
                            if (g + 1 < NG) {
#pragma unroll
                                for (int jj = 0; jj < NCO; ++jj) a_nxt[jj] = frag_a(g + 1, jj);
#pragma unroll
                                for (int n2 = 0; n2 < NPT; ++n2)
#pragma unroll
                                    for (int s2 = 0; s2 < 4; ++s2) b_nxt[n2][s2] = frag_b(g + 1, n2, s2);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < NG) {
#pragma unroll
                for (int j = 0; j < NCO; ++j) a_cur[j] = a_nxt[j];
#pragma unroll
                for (int nt = 0; nt < NPT; ++nt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) b_cur[nt][s] = b_nxt[nt][s];
            }
        }
    }

    // epilogue: lane = pixel (coalesced), register = output channel
#undef VF_LOAD_CHUNK
#undef VF_STORE_CHUNK
#ifdef VF_CONV_STAMPS
    st_[2] = clock64();
#endif
    if constexpr (MODE == 4) {        // dx (S, Cout, 2H, 2W): pixel (i, j) of the tile -> the 2x2 block (2i+a, 2j+b)
#pragma unroll
        for (int nt = 0; nt < NPT; ++nt) {
            const int p = pw * 32 * NPT + nt * 32 + li;
            const int s = G::IM > 1 ? s0 + (p >> (2 * LOGW)) : s0;
            const int pix = G::IM > 1 ? (p & (G::HW - 1)) : r0 * G::W + p;
            if (s >= a.S) continue;
            const int pi = pix >> LOGW, pj = pix & (G::W - 1);
            const int cob = co0 + cw * 32 + 4 * lh;
            const size_t oo = ((size_t)s * a.Cout + cob) * (4 * G::HW) + (size_t)(2 * pi) * (2 * G::W) + 2 * pj;
            float* ob = a.y + oo;
            // residual (same layout as dx): the gradient that reaches this conv's INPUT through its other consumer
            // (the decoder's skip connection), added here instead of by an autograd add kernel
            float2 ra[16], rb[16];
            if (a.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dco = (r & 3) + 8 * (r >> 2);
                    if (cob + dco < a.Cout) {
                        const float* q = a.res + oo + (size_t)dco * (4 * G::HW);
                        ra[r] = *reinterpret_cast<const float2*>(q);
                        rb[r] = *reinterpret_cast<const float2*>(q + 2 * G::W);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = (r & 3) + 8 * (r >> 2);
                if (cob + dco < a.Cout) {
                    float* o = ob + (size_t)dco * (4 * G::HW);
                    float2 v0 = make_float2(acc[0][nt][r], acc[1][nt][r]), v1 = make_float2(acc[2][nt][r], acc[3][nt][r]);
                    if (a.res) { v0.x += ra[r].x; v0.y += ra[r].y; v1.x += rb[r].x; v1.y += rb[r].y; }
                    *reinterpret_cast<float2*>(o) = v0;
                    *reinterpret_cast<float2*>(o + 2 * G::W) = v1;
                }
            }
        }
        return;
    }
    // All loads of a 32x32 tile (residual, biases) are issued before any of its stores so that
    // their latency overlaps instead of forming a load->add->store chain per element.
#pragma unroll
    for (int j = 0; j < NCO; ++j)
#pragma unroll
    for (int nt = 0; nt < NPT; ++nt) {
        const int p = pw * 32 * NPT + nt * 32 + li;
        const int s = G::IM > 1 ? s0 + (p >> (2 * LOGW)) : s0;
        const int pix = G::IM > 1 ? (p & (G::HW - 1)) : r0 * G::W + p;
        if (s >= a.S || j >= nco_here) continue;
        const int cob = co0 + j * TCO + cw * 32 + 4 * lh;
        // split output (dgrad of a conv on concatenated inputs): whole 64-channel tiles go to y or to y2
        const bool second = a.y2 && co0 + j * TCO >= a.C1o;
        float* const yout = second ? a.y2 : a.y;
        const size_t ob = second ? ((size_t)s * (a.Cout - a.C1o) + (cob - a.C1o)) * G::HW + pix
                                 : ((size_t)s * (a.y2 ? a.C1o : a.Cout) + cob) * G::HW + pix;
        if (a.ksplit > 1) {            // raw partial sums; bias / residual are added by the reduce kernel
            float* wsp = a.ws + (size_t)split * a.S * a.Cout * G::HW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = (r & 3) + 8 * (r >> 2);
                if (cob + dco < a.Cout) wsp[ob + (size_t)dco * G::HW] = acc[j][nt][r];
            }
            continue;
        }
        // one uniform branch per operand kind with its 16 loads back to back (channel index clamped instead of
        // predicated): a branch between two loads makes the compiler wait for the first before the second
        float add[16], ad2[16], ad3[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = ad2[r] = ad3[r] = 0.f;
        if (a.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = min((r & 3) + 8 * (r >> 2), a.Cout - 1 - cob);
                add[r] = a.res[ob + (size_t)dco * G::HW];
            }
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ad2[r] = a.bias[min(cob + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
        }
        if (a.vbias) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ad3[r] = a.vbias[(size_t)s * a.Cout + min(cob + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] += ad2[r] + ad3[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dco = (r & 3) + 8 * (r >> 2);
            if (cob + dco < a.Cout) yout[ob + (size_t)dco * G::HW] = acc[j][nt][r] + add[r];
        }
    }
#ifdef VF_CONV_STAMPS
    __syncthreads();
    if (tid == 0 && a.ws) {
        st_[3] = clock64();
        rt_[1] = wall_clock64();
        long long* o = reinterpret_cast<long long*>(a.ws) + (size_t)blockIdx.x * 8;
        o[0] = st_[0]; o[1] = st_[1]; o[2] = st_[2]; o[3] = st_[3]; o[4] = rt_[0]; o[5] = rt_[1];
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        o[6] = hwid;
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// weight gradient
struct WgradArgs {
    const float* x;
    const float* dy;
    float* ws;       // [slab][tap][CoutP][CinQ]
    int S, Cin, Cout, CoutP, CinQ;   // CoutP multiple of 64, CinQ multiple of 32
    int ntiles, tiles_per_slice;
    const float* x2;                 // 1x1 on a channel concatenation: input channels [C1, Cin) live here
    int C1;
};

// BIG = true : 64 co x 64 ci tile, waves = 2 (co) x 2 (ci), ONE workgroup per CU with the whole
//              512-register file per lane: 144 accumulators + a full pixel tile prefetched in
//              registers while the previous one is multiplied (stride-1 / upsampled inputs).
// BIG = false: 64 co x 32 ci tile, waves = 2 (co) x 2 (pixel halves, summed through LDS), two
//              workgroups per CU, no register prefetch (stride-2 inputs: the patch is too large).
template <int KS, int LOGW, int MODE, bool BIG>
__global__ __launch_bounds__(256, BIG ? 1 : 2) void conv_wgrad_kernel(WgradArgs a) {
    constexpr int TPIX = 128;
    using G = Geo<KS, LOGW, MODE, TPIX>;
    constexpr int NT = KS * KS;
    constexpr int TCI = BIG ? 64 : 32;
    constexpr int DYS = TPIX + 4;                 // 4*odd row stride: conflict-free ds_read_b128 down a column
    constexpr int PSO = G::PS | 1;                // odd plane stride
    constexpr int NX4 = TCI * G::IM * G::PH * G::Q;
    constexpr int ND4 = TCO * (TPIX / 4);
    constexpr int OHW = G::HW;

    constexpr int LDSF = TCO * DYS + TCI * PSO;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];   // dY tile | input patch | k-half reduce
    float* const dyl = lds;
    float* const xl = lds + TCO * DYS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1;
    const int kq = BIG ? 0 : (wid >> 1);          // pixel half of this wave (small tile only)
    const int ciw = BIG ? (wid >> 1) : 0;         // ci half of this wave (big tile only)
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware decode (see wino_wgrad_kernel)
    const unsigned lgc = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                   gridDim.x * gridDim.y * gridDim.z);
    const int bx = lgc % gridDim.x, by = (lgc / gridDim.x) % gridDim.y, bz = lgc / (gridDim.x * gridDim.y);
    const int co0 = bx * TCO, ci0 = by * TCI;
    const int t_begin = bz * a.tiles_per_slice;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_slice);

    for (int i = tid; i < TCI * PSO; i += 256) xl[i] = 0.f;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};

    const float* ab = dyl + (cw * 32 + li) * DYS + kq * 64 + 4 * lh;
    const float* bb = xl + (ciw * 32 + li) * PSO + G::pix_off(kq * 64 + 4 * lh);

    // Global -> register staging of one pixel tile (issued a whole tile ahead of its use when the
    // register budget allows: PF) and register -> LDS write-out.
    constexpr int NXF = NX4 / 256;
    constexpr bool XT = (NX4 % 256) != 0;
    constexpr bool PF = BIG;
    float4 dreg[ND4 / 256], xreg[NXF > 0 ? NXF : 1];
    float4 xtail = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_dy = [&](int e, int tile) -> float4 {
        const int s0 = G::IM > 1 ? tile * G::IM : tile / G::TPI;
        const int r0 = G::IM > 1 ? 0 : (tile % G::TPI) * G::TH;
        const int q = e & 31, co = e >> 5;                   // 32 float4 per row
        const int p = 4 * q;
        const int s = G::IM > 1 ? s0 + (p >> (2 * LOGW)) : s0;
        const int pix = G::IM > 1 ? (p & (OHW - 1)) : r0 * G::W + p;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s < a.S && co0 + co < a.Cout)
            v = *reinterpret_cast<const float4*>(a.dy + ((size_t)s * a.Cout + co0 + co) * OHW + pix);
        return v;
    };
    auto load_x = [&](int e, int tile) -> float4 {
        const int s0 = G::IM > 1 ? tile * G::IM : tile / G::TPI;
        const int r0 = G::IM > 1 ? 0 : (tile % G::TPI) * G::TH;
        const int q = e % G::Q;
        const int t1 = e / G::Q;
        const int pr = t1 % G::PH;
        const int t2 = t1 / G::PH;
        const int im = t2 % G::IM, ci = t2 / G::IM;
        if (KS == 1 && a.x2) {                                    // concatenated input: tile-uniform source
            const bool second = ci0 >= a.C1;
            return load_patch4<G, MODE>(second ? a.x2 : a.x, a.S, second ? a.Cin - a.C1 : a.C1, s0 + im,
                                        second ? ci0 - a.C1 + ci : ci0 + ci, r0, pr, q);
        }
        return load_patch4<G, MODE>(a.x, a.S, a.Cin, s0 + im, ci0 + ci, r0, pr, q);
    };
    auto store_x = [&](int e, const float4& v) {
        const int q = e % G::Q;
        const int t1 = e / G::Q;                              // (ci*IM + im)*PH + pr
        const int ci = t1 / (G::IM * G::PH), rest = t1 % (G::IM * G::PH);
        float* d = xl + ci * PSO + rest * G::PW + 4 * q + (KS == 1 ? 0 : 4);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    };
#define VF_WG_LOAD(TILE)                                                                    \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < ND4 / 256; ++i) dreg[i] = load_dy(tid + i * 256, (TILE)); \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) xreg[i] = load_x(tid + i * 256, (TILE));        \
        if (XT && tid + NXF * 256 < NX4) xtail = load_x(tid + NXF * 256, (TILE));            \
    }
#define VF_WG_STORE()                                                                       \
    {                                                                                       \
        _Pragma("unroll") for (int i = 0; i < ND4 / 256; ++i) {                              \
            const int e = tid + i * 256;                                                    \
            *reinterpret_cast<float4*>(dyl + (e >> 5) * DYS + 4 * (e & 31)) = dreg[i];       \
        }                                                                                   \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) store_x(tid + i * 256, xreg[i]);     \
        if (XT && tid + NXF * 256 < NX4) store_x(tid + NXF * 256, xtail);                    \
    }

    if (PF && t_begin < t_end) VF_WG_LOAD(t_begin);
    for (int tile = t_begin; tile < t_end; ++tile) {
        __syncthreads();
        if (!PF) VF_WG_LOAD(tile);
        VF_WG_STORE();
        __syncthreads();
        if (PF && tile + 1 < t_end) VF_WG_LOAD(tile + 1);
        // 8 pixel groups of 8 per wave; k order inside a group: MFMA step e pairs pixel e (lane
        // half 0) with pixel 4+e (half 1) -> the four A values of a lane are one ds_read_b128.
        // Software pipeline: the B fragments of stage (group, tap row) + 1 are fetched before the
        // MFMAs of the current stage are issued.
        constexpr int NSTG = (BIG ? 16 : 8) * KS;          // stages per chunk: (pixel group of 8, kh)
        auto frag_b = [&](int stg, int kw, int e) -> float {
            const int grp = stg / KS, kh = stg % KS;
            const int d = G::pix_off(8 * grp) - G::pix_off(0);
            return bb[d + (MODE == 1 ? 2 * e : e) + kh * G::PW + kw];
        };
        float b_cur[KS][4];
#pragma unroll
        for (int kw = 0; kw < KS; ++kw)
#pragma unroll
            for (int e = 0; e < 4; ++e) b_cur[kw][e] = frag_b(0, kw, e);
        float4 a4 = *reinterpret_cast<const float4*>(ab);
#pragma unroll
        for (int stg = 0; stg < NSTG; ++stg) {
            float b_nxt[KS][4];
            float4 a_nxt = a4;
            if (stg + 1 < NSTG) {
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                    for (int e = 0; e < 4; ++e) b_nxt[kw][e] = frag_b(stg + 1, kw, e);
                if ((stg + 1) % KS == 0) a_nxt = *reinterpret_cast<const float4*>(ab + 8 * ((stg + 1) / KS));
            }
            __builtin_amdgcn_sched_barrier(0);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    const int tap = (stg % KS) * KS + kw;
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], b_cur[kw][e], acc[tap], 0, 0, 0);
                }
            __builtin_amdgcn_sched_barrier(0);
            if (stg + 1 < NSTG) {
                a4 = a_nxt;
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                    for (int e = 0; e < 4; ++e) b_cur[kw][e] = b_nxt[kw][e];
            }
        }
    }

#undef VF_WG_LOAD
#undef VF_WG_STORE
    // the two pixel halves (kq) of each co-half are summed through LDS, in passes of NTP taps
    constexpr int NTP = NT > 1 ? (NT + 1) / 2 : 1;
    static_assert(2 * NTP * 16 * 64 <= LDSF, "k-half reduce buffer does not fit");
#pragma unroll
    for (int t0 = 0; t0 < (BIG ? 0 : NT); t0 += NTP) {
        __syncthreads();
        if (kq == 1) {
#pragma unroll
            for (int t = t0; t < (t0 + NTP < NT ? t0 + NTP : NT); ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) lds[((cw * NTP + (t - t0)) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (kq == 0) {
#pragma unroll
            for (int t = t0; t < (t0 + NTP < NT ? t0 + NTP : NT); ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += lds[((cw * NTP + (t - t0)) * 16 + r) * 64 + lane];
        }
    }
    if (kq != 0) return;
    const int slab = bz;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int ci = ci0 + ciw * 32 + li;
            if (ci < a.CinQ) a.ws[(((size_t)slab * NT + tap) * a.CoutP + co) * a.CinQ + ci] = acc[tap][r];
        }
    }
}

// dw[co][ci][tap] = sum_slab ws[slab][tap][co][ci]: wgrad_reduce_body (wgrad_reduce.h), one launch per layer here; a host
// that can postpone the weight gradients runs the *_main entry points instead and sums every layer's slabs with ONE
// vf_wino44_reduce_multi launch at the end of the backward pass (round 6: 38 launches of ~6 us per training iteration gone).
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                           int nslab, int NT, int Cout, int Cin, int CoutP,
                                                           int CinQ) {
    wgrad_reduce_body(ws, dw, nslab, NT, Cout, Cin, CoutP, CinQ, (int)blockIdx.x);
}

// ----------------------------------------------------------------------------------------------
// 1x1 weight gradient: dW[co][ci] = sum_{s,p} dY[s][co][p] X[s][ci][p] -- a plain GEMM with K = S*H*W.
// A 64x64 output tile loads 2 bytes per MAC-row and is bound by the per-CU load path (measured 45-69 TF,
// ~4 TB/s of L2->LDS traffic); this kernel uses 128 co x 128 ci tiles (half the bytes per flop): four waves,
// each 64 x 64 outputs = 2x2 MFMA blocks, K in steps of 64 pixels, both operands read from [channel][pixel]
// LDS images with ds_read_b128 (k order inside 8 pixels: MFMA step e pairs pixel e | 4+e), next step's global
// loads in registers while the current one is multiplied, two workgroups per CU.  Split-K over pixel steps into
// slabs summed by wgrad_reduce_kernel in fixed order.  x may be the never-materialised concatenation [x | x2].
struct Wgrad1Args {
    const float* x;
    const float* x2;
    const float* dy;
    float* ws;                    // [slab][CoutP][CinQ]
    int S, Cin, Cout, CoutP, CinQ, C1, HW;
    int nsteps, steps_per_slice;
};

// TM x TN = output channels x input channels per workgroup (64 or 128 each): waves 2 x 2, each (TM/2) x (TN/2)
template <int TM, int TN>
__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_kernel(Wgrad1Args a) {
    constexpr int KP = 64, RS = KP + 4;                 // pixels per step, LDS row stride (4*odd: conflict-free b128)
    constexpr int NI = TM / 64, NJ = TN / 64;           // 32x32 MFMA blocks per wave
    constexpr int ND = TM * 16 / 256, NX = TN * 16 / 256;   // float4 per thread per step (4 or 8 each)
    __shared__ __attribute__((aligned(16))) float Dl[TM * RS];
    __shared__ __attribute__((aligned(16))) float Xl[TN * RS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, ciw = wid >> 1, li = lane & 31, lh = lane >> 5;
    // XCD-aware decode (see wino_wgrad_kernel): the (co, ci) tile pairs of one K slice read the same x / dY pixels;
    // consecutive logical ids share an XCD, so those pixels come from HBM once per slice
    const unsigned lgc = xcd_remap(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                                   gridDim.x * gridDim.y * gridDim.z);
    const int bx = lgc % gridDim.x, by = (lgc / gridDim.x) % gridDim.y, bz = lgc / (gridDim.x * gridDim.y);
    const int co0 = bx * TM, ci0 = by * TN;
    const int t_begin = bz * a.steps_per_slice;
    const int t_end = min(a.nsteps, t_begin + a.steps_per_slice);
    const int spv = a.HW / KP;                            // steps per view

    // staging: row = channel, 16 float4 per row.  Rows beyond the tensor re-read its last channel: their
    // products land in slab rows / columns that are never reduced.
    const int srow = tid >> 4, sq = tid & 15;
    size_t doff[8], xoff[8];
    const float* xb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = srow + 16 * i;
        doff[i] = (size_t)min(co0 + row, a.Cout - 1) * a.HW + 4 * sq;
        const int ci = min(ci0 + row, a.Cin - 1);
        const bool second = a.x2 && ci >= a.C1;
        xb[i] = second ? a.x2 : a.x;
        xoff[i] = (size_t)(second ? ci - a.C1 : ci) * a.HW + 4 * sq;
    }
    const int xC = a.x2 ? a.C1 : a.Cin, x2C = a.Cin - a.C1;
    float4 dr0, dr1, dr2, dr3, dr4, dr5, dr6, dr7, xr0, xr1, xr2, xr3, xr4, xr5, xr6, xr7;
    dr0 = dr1 = dr2 = dr3 = dr4 = dr5 = dr6 = dr7 = xr0 = xr1 = xr2 = xr3 = xr4 = xr5 = xr6 = xr7 =
        make_float4(0.f, 0.f, 0.f, 0.f);
#define VF_W1_LD(I, T_)                                                                                   \
    {                                                                                                     \
        const int sv = (T_) / spv, p0 = ((T_) % spv) * KP;                                                 \
        if constexpr ((I) < ND)                                                                           \
            dr##I = *reinterpret_cast<const float4*>(a.dy + (size_t)sv * a.Cout * a.HW + doff[I] + p0);    \
        if constexpr ((I) < NX)                                                                           \
            xr##I = *reinterpret_cast<const float4*>(xb[I] + (size_t)sv * (xb[I] == a.x ? xC : x2C) * a.HW + xoff[I] + p0); \
    }
#define VF_W1_ST(I)                                                                                       \
    {                                                                                                     \
        if constexpr ((I) < ND) *reinterpret_cast<float4*>(Dl + (srow + 16 * (I)) * RS + 4 * sq) = dr##I;   \
        if constexpr ((I) < NX) *reinterpret_cast<float4*>(Xl + (srow + 16 * (I)) * RS + 4 * sq) = xr##I;   \
    }
#define VF_W1_LOAD(T_) { VF_W1_LD(0, T_) VF_W1_LD(1, T_) VF_W1_LD(2, T_) VF_W1_LD(3, T_) VF_W1_LD(4, T_) VF_W1_LD(5, T_) VF_W1_LD(6, T_) VF_W1_LD(7, T_) }
#define VF_W1_STORE() { VF_W1_ST(0) VF_W1_ST(1) VF_W1_ST(2) VF_W1_ST(3) VF_W1_ST(4) VF_W1_ST(5) VF_W1_ST(6) VF_W1_ST(7) }

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x16){0};
    const float* ab = Dl + (cw * (TM / 2) + li) * RS + 4 * lh;
    const float* bb = Xl + (ciw * (TN / 2) + li) * RS + 4 * lh;

    if (t_begin < t_end) {
        VF_W1_LOAD(t_begin);
        for (int t = t_begin; t < t_end; ++t) {
            __syncthreads();
            VF_W1_STORE();
            __syncthreads();
            VF_W1_LOAD(min(t + 1, t_end - 1));            // unconditional (clamped): the loads stay countable
            float4 fa[NI], fb[NJ];
#pragma unroll
            for (int i = 0; i < NI; ++i) fa[i] = *reinterpret_cast<const float4*>(ab + 32 * i * RS);
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const float4*>(bb + 32 * j * RS);
#pragma unroll
            for (int g = 0; g < KP / 8; ++g) {
                float4 na[NI], nb[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) na[i] = fa[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) nb[j] = fb[j];
                if (g + 1 < KP / 8) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) na[i] = *reinterpret_cast<const float4*>(ab + 32 * i * RS + 8 * (g + 1));
#pragma unroll
                    for (int j = 0; j < NJ; ++j) nb[j] = *reinterpret_cast<const float4*>(bb + 32 * j * RS + 8 * (g + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = na[i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = nb[j];
            }
        }
    }
#undef VF_W1_LD
#undef VF_W1_ST
#undef VF_W1_LOAD
#undef VF_W1_STORE
    float* sb = a.ws + (size_t)bz * a.CoutP * a.CinQ;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ci = ci0 + ciw * (TN / 2) + j * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cw * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                sb[(size_t)co * a.CinQ + ci] = acc[i][j][r];
            }
        }
}

// OIHW -> packed forward  [co tile][ci chunk][group][co 64][ci 8]   (M = Cout, K = Cin x taps)
//      and packed backward [ci tile][co chunk][group][ci 64][co 8]   (dgrad: M = Cin, K = Cout x flipped taps)
// group = tap for 3x3 (chunk = 8 channels), = 8-channel sub-chunk for 1x1 (chunk = 32 channels).
// Zero padded to whole tiles / chunks.
__device__ __forceinline__ void pack_one(const float* __restrict__ w, float* __restrict__ wf,
                                         float* __restrict__ wb, int Cout, int Cin, int KS, size_t nf, size_t nb,
                                         size_t idx) {
    const int NT = KS * KS, CK = KS == 3 ? 8 : 32, NG = KS == 3 ? 9 : 4;
    const bool bwd = idx >= nf;
    if (bwd) {
        idx -= nf;
        if (idx >= nb || !wb) return;
    }
    const int M = bwd ? Cin : Cout, K = bwd ? Cout : Cin;      // M: tile dim, K: chunk dim
    const int nchunk = (K + CK - 1) / CK;
    const int k8 = idx & 7;
    const int m = (idx >> 3) & 63;
    size_t t = idx >> 9;
    const int g = t % NG;
    t /= NG;
    const int chunk = t % nchunk;
    const int mt = t / nchunk;
    const int mm = mt * 64 + m;
    const int kk = chunk * CK + (KS == 3 ? k8 : g * 8 + k8);
    const int tap = KS == 3 ? (bwd ? NT - 1 - g : g) : 0;
    float v = 0.f;
    if (mm < M && kk < K) {
        const int co = bwd ? kk : mm, ci = bwd ? mm : kk;
        v = w[((size_t)co * Cin + ci) * NT + tap];
    }
    (bwd ? wb : wf)[idx] = v;
}

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb,
                                    int Cout, int Cin, int KS, size_t nf, size_t nb) {
    pack_one(w, wf, wb, Cout, Cin, KS, nf, nb, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

// All conv layers of a network in ONE launch.  desc[l] = {w, wf, wb, Cout, Cin, KS, nf, nb, first
// block}; a block finds its layer by binary search over the first-block column.
struct PackDesc {
    const float* w;
    float* wf;
    float* wb;
    long long Cout, Cin, KS, nf, nb, first_block;
};
__global__ void pack_weights_multi_kernel(const PackDesc* __restrict__ desc, int nlayers) {
    int lo = 0, hi = nlayers;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid;
    }
    const PackDesc d = desc[lo];
    const size_t idx = ((size_t)blockIdx.x - (size_t)d.first_block) * blockDim.x + threadIdx.x;
    pack_one(d.w, d.wf, d.wb, (int)d.Cout, (int)d.Cin, (int)d.KS, (size_t)d.nf, (size_t)d.nb, idx);
}

// y[s][c][h][w] = sum of the 2x2 block of x (backward of nearest x2 upsampling)
__global__ void sumpool2_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n_out, int Wo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const size_t row = i / Wo;            // (s*C + c)*Ho + h
    const int w = i - row * Wo;
    const float* p = x + row * 2 * (2 * Wo) + 2 * w;
    y[i] = (p[0] + p[1]) + (p[2 * Wo] + p[2 * Wo + 1]);
}

// y = sum_split ws[split] + bias[co] + view_bias[s][co] + residual      (float4 over S*Cout*HW)
__global__ void conv_splitk_reduce_kernel(const float4* __restrict__ ws, const float* __restrict__ bias,
                                          const float* __restrict__ vbias, const float4* __restrict__ res,
                                          float4* __restrict__ y, int ksplit, size_t n4, int HW4, int Cout) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    // every operand is requested before the first add (ksplit <= 16: two batches of eight independent loads on
    // clamped indices; a plain accumulate loop pays one memory round trip per partial -- this kernel runs ~100 times
    // per reverse step of the sampler).  Fixed summation order.
    const size_t sc = i / HW4;                 // s*Cout + co
    float b = 0.f;
    if (bias) b += bias[sc % Cout];
    if (vbias) b += vbias[sc];
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (res) r = res[i];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < ksplit; k0 += 8) {
        float4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = ws[(size_t)min(k0 + j, ksplit - 1) * n4 + i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (k0 + j < ksplit) { v.x += t[j].x; v.y += t[j].y; v.z += t[j].z; v.w += t[j].w; }
        }
    }
    y[i] = make_float4(v.x + r.x + b, v.y + r.y + b, v.z + r.z + b, v.w + r.w + b);
}

// Split-K reduce + epilogue + GroupNorm(+Swish) in ONE launch (sampler regime, vf_conv_fwd_gn): one workgroup per
// (view, group) sums the K-split partials of its channels, adds bias / per-view bias / residual, optionally stores the
// conv output, and -- the whole group being in its registers -- normalises it (two-pass mean / variance as
// gn_fwd_kernel) into `a_out`.  Replaces conv_splitk_reduce_kernel + gn_fwd_kernel.
template <int NV, int NT>
__global__ __launch_bounds__(NT) void conv_splitk_reduce_gn_kernel(const float4* __restrict__ ws, const float* __restrict__ bias,
                                                                    const float* __restrict__ vbias,
                                                                    const float4* __restrict__ res, float4* __restrict__ y,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, float4* __restrict__ a_out,
                                                                    int ksplit, size_t n4_total, int hw4sh, int Cout, int cpg,
                                                                    float eps, int silu) {
    __shared__ float red[NT / 64];
    constexpr int KB = NV >= 8 ? 1 : 8 / NV;                 // partials per batch: >= 8 independent loads in flight
    const int G = Cout / cpg;
    const int sg = blockIdx.x, s = sg / G, g = sg - s * G;
    const int n4 = cpg << hw4sh;
    const size_t base4 = ((size_t)s * Cout + (size_t)g * cpg) << hw4sh;
    float4 v[NV];
    size_t gi[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        gi[i] = base4 + min((int)threadIdx.x + i * NT, n4 - 1);
    }
    for (int k0 = 0; k0 < ksplit; k0 += KB) {
        float4 t[KB][NV];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
#pragma unroll
            for (int i = 0; i < NV; ++i) t[kk][i] = ws[(size_t)min(k0 + kk, ksplit - 1) * n4_total + gi[i]];
#pragma unroll
        for (int kk = 0; kk < KB; ++kk)
            if (k0 + kk < ksplit) {
#pragma unroll
                for (int i = 0; i < NV; ++i) { v[i].x += t[kk][i].x; v[i].y += t[kk][i].y; v[i].z += t[kk][i].z; v[i].w += t[kk][i].w; }
            }
    }
    float gam[NV], bet[NV];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int idx = min((int)threadIdx.x + i * NT, n4 - 1);
        const int c = g * cpg + (idx >> hw4sh);
        float b = 0.f;
        if (bias) b += bias[c];
        if (vbias) b += vbias[(size_t)s * Cout + c];
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (res) r = res[gi[i]];
        v[i] = make_float4(v[i].x + r.x + b, v[i].y + r.y + b, v[i].z + r.z + b, v[i].w + r.w + b);
        gam[i] = gamma[c];
        bet[i] = beta[c];
        const bool ok = (int)threadIdx.x + i * NT < n4;
        if (y && ok) y[gi[i]] = v[i];
        if (!ok) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float inv_n = 1.0f / (float)(n4 * 4);
    const float mean = block_sum<NT>(sum, red) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        const float q = (a * a + b * b) + (c * c + d * d);
        sq += (int)threadIdx.x + i * NT < n4 ? q : 0.f;
    }
    const float var = block_sum<NT>(sq, red) * inv_n;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((int)threadIdx.x + i * NT < n4) {
            const float ga = gam[i] * rstd;
            const float be = bet[i] - mean * ga;
            float4 o = make_float4(v[i].x * ga + be, v[i].y * ga + be, v[i].z * ga + be, v[i].w * ga + be);
            if (silu) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
            a_out[gi[i]] = o;
        }
    }
}

// launches the fused reduce + GroupNorm for a (view, group) of n4g float4; false if the group is too large
inline bool launch_reduce_gn(const ConvArgs& a, int ks, size_t n4_total, int HW, hipStream_t st) {
    const int cpg = a.Cout / a.gn_groups;
    const long n4g = (long)cpg * HW / 4;
    int hw4sh = 0;
    while ((1 << hw4sh) < HW / 4) ++hw4sh;
    const dim3 grid(a.S * a.gn_groups);
    float4* y4 = a.gn_store_y ? (float4*)a.y : nullptr;
#define VF_RGN(NV, NT)                                                                                          \
    {                                                                                                           \
        hipLaunchKernelGGL((conv_splitk_reduce_gn_kernel<NV, NT>), grid, dim3(NT), 0, st, (const float4*)a.ws, a.bias, \
                           a.vbias, (const float4*)a.res, y4, a.gn_gamma, a.gn_beta, (float4*)a.gn_out, ks, n4_total, \
                           hw4sh, a.Cout, cpg, a.gn_eps, a.gn_silu);                                            \
        return true;                                                                                            \
    }
    if (n4g <= 256) VF_RGN(1, 256)
    if (n4g <= 512) VF_RGN(2, 256)
    if (n4g <= 1024) VF_RGN(4, 256)
    if (n4g <= 2048) VF_RGN(8, 256)
    if (n4g <= 4096) VF_RGN(8, 512)
    if (n4g <= 8192) VF_RGN(8, 1024)
#undef VF_RGN
    return false;
}

extern "C" int vf_gn_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int S,
                         int C, int HW, int groups, float eps, int silu, void* stream);

// Split-K factor: only when the natural grid cannot fill the chip (small S: the sampler).  Measured over the
// sampler's layer shapes at S = 1 / 6 / 12 (tools/small_conv.py): the fastest split is the largest one that still
// gives every workgroup a CU of its own in a single round (nblk * k <= 256) -- a second round, or more partial
// sums than that for the reduce kernel to add up (k > 16), costs more than the shorter K loop saves.
inline int choose_ksplit(int nblk, int nchunks) {
    static const int budget = [] { const char* e = getenv("VF_CONV_KSPLIT_WGS"); return e ? atoi(e) : 256; }();
    if (nblk >= budget || nchunks < 2) return 1;
    int k = budget / nblk;
    if (k > 16) k = 16;
    if (k > nchunks) k = nchunks;
    return k < 1 ? 1 : k;
}

template <int KS, int LOGW, int MODE, int NPT, int NCO = 1>
int conv_blocks(const ConvArgs& a) {
    using G = Geo<KS, LOGW, MODE, 64 * NPT>;
    const int ntiles = G::IM > 1 ? (a.S + G::IM - 1) / G::IM : a.S * G::TPI;
    return ntiles * ((a.CoutP / TCO + NCO - 1) / NCO);
}

template <int KS, int LOGW, int MODE, int NPT, int NCO = 1>
int launch_conv_npt(ConvArgs a, hipStream_t st, long ws_floats) {
    constexpr int HW = 1 << (2 * LOGW);
    const int nblk = conv_blocks<KS, LOGW, MODE, NPT, NCO>(a);
    const size_t out_floats = (size_t)a.S * a.Cout * HW;
    int ks = choose_ksplit(nblk, a.CinP / (KS == 3 ? 8 : 32));
    if (a.ws == nullptr) ks = 1;
    while (ks > 1 && (size_t)ks * out_floats > (size_t)ws_floats) --ks;
    a.ksplit = ks;
    using G = Geo<KS, LOGW, MODE, 64 * NPT>;
    // every patch element exists (see load_patch4): whole 32-channel chunks from either source, whole view groups
    const bool safe = KS == 1 && a.Cin % 32 == 0 && (!a.x2 || a.C1 % 32 == 0) && (G::IM == 1 || a.S % G::IM == 0);
    if (KS == 1 && safe)
        hipLaunchKernelGGL((conv_mfma_kernel<KS, LOGW, MODE, NPT, NCO, KS == 1>), dim3(nblk * ks), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<KS, LOGW, MODE, NPT, NCO, false>), dim3(nblk * ks), dim3(256), 0, st, a);
    bool gn_done = false;
    if (ks > 1) {
        const size_t n4 = out_floats / 4;
        if (a.gn_out) gn_done = launch_reduce_gn(a, ks, n4, HW, st);
        if (!gn_done)
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st,
                               (const float4*)a.ws, a.bias, a.vbias, (const float4*)a.res, (float4*)a.y, ks, n4, HW / 4,
                               a.Cout);
    }
    if (a.gn_out && !gn_done)         // the conv wrote y itself (no split-K here): plain GroupNorm launch behind it
        return vf_gn_fwd(a.y, a.gn_gamma, a.gn_beta, a.gn_out, a.gn_stats, a.gn_stats + (size_t)a.S * a.gn_groups, a.S,
                         a.Cout, HW, a.gn_groups, a.gn_eps, a.gn_silu, (void*)st);
    VF_RETURN_LAST_ERROR();
}

// Tile-size choice (measured, profiles/r01_conv_tile_sweep.txt): 128-pixel tiles (NPT=2) are the
// fastest wherever they fill the chip; 256-pixel tiles lose occupancy and are never faster; small
// feature maps (16x16, 8x8 at S=96) fall back to 64-pixel tiles so that the 256 CUs stay filled
// and load-balanced.
constexpr int CONV_MIN_WGS = 3 * 256;

inline int npt_override() {
    static const int v = [] {
        const char* e = getenv("VF_CONV_NPT");       // tuning aid: force the pixel-tile size
        return e ? atoi(e) : 0;
    }();
    return v;
}

// 1x1 convs.  (A 64*NCO-channel workgroup tile -- NCO = 2, 3 instantiations of conv_mfma_kernel -- was measured:
// 5-20 % faster on a few 16x16 / 8x8 shapes, 10-100 % slower wherever the grid stops filling the chip, net worse;
// only NCO = 1 is compiled.)
template <int LOGW, int NPT>
int launch_conv_1x1(const ConvArgs& a, hipStream_t st, long ws_floats) {
    return launch_conv_npt<1, LOGW, 0, NPT, 1>(a, st, ws_floats);
}

template <int KS, int LOGW, int MODE>
int launch_conv(const ConvArgs& a, hipStream_t st, long ws_floats) {
    constexpr bool ok1 = LOGW <= 4;                       // 64-pixel tiles
    if constexpr (KS == 1) {
        if constexpr (ok1) {
            const int force = npt_override();
            if (force == 1 || (force == 0 && conv_blocks<1, LOGW, 0, 2>(a) < CONV_MIN_WGS))
                return launch_conv_1x1<LOGW, 1>(a, st, ws_floats);
        }
        return launch_conv_1x1<LOGW, 2>(a, st, ws_floats);
    }
    if constexpr (ok1) {
        const int force = npt_override();
        if (force == 1 || (force == 0 && conv_blocks<KS, LOGW, MODE, 2>(a) < CONV_MIN_WGS))
            return launch_conv_npt<KS, LOGW, MODE, 1>(a, st, ws_floats);
    }
    return launch_conv_npt<KS, LOGW, MODE, 2>(a, st, ws_floats);
}

template <int KS, int LOGW, int MODE>
int launch_wgrad(WgradArgs a, float* dw, size_t ws_floats, hipStream_t st, long long* desc = nullptr, int* nblocks = nullptr) {
    using G = Geo<KS, LOGW, MODE, 128>;
    constexpr int NT = KS * KS;
    constexpr bool BIG = MODE != 1;                  // stride-2 patches do not fit the big tile
    constexpr int TCI = BIG ? 64 : 32;
    a.ntiles = G::IM > 1 ? (a.S + G::IM - 1) / G::IM : a.S * G::TPI;
    const int nco = a.CoutP / TCO, nci = (a.CinQ + TCI - 1) / TCI;
    const size_t slab_floats = (size_t)NT * a.CoutP * a.CinQ;
    int z = (BIG ? 256 : WGRAD_TARGET_WGS) / (nco * nci);   // never more workgroups than fit at once
    if (z < 1) z = 1;
    if (z > a.ntiles) z = a.ntiles;
    const size_t zmax = ws_floats / slab_floats;
    if (zmax < 1) return (int)hipErrorInvalidValue;
    if ((size_t)z > zmax) z = (int)zmax;
    a.tiles_per_slice = (a.ntiles + z - 1) / z;
    z = (a.ntiles + a.tiles_per_slice - 1) / a.tiles_per_slice;
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, LOGW, MODE, BIG>), dim3(nco, nci, z), dim3(256), 0, st, a);
    const int total = NT * a.Cout * a.Cin;
    if (desc) {                                       // the slab sum joins the caller's deferred multi launch
        *nblocks = wgrad_reduce_row(desc, a.ws, dw, z, NT, a.Cout, a.Cin, a.CoutP, a.CinQ);
        VF_RETURN_LAST_ERROR();
    }
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st, a.ws, dw, z, NT,
                       a.Cout, a.Cin, a.CoutP, a.CinQ);
    VF_RETURN_LAST_ERROR();
}

// 1x1 wgrad tile side for a channel count: 128 unless the padding to 128 costs more than 20 %
inline int round_up(int v, int m);
inline int wgrad1_tile(int C) { return 5 * ((C + 127) / 128 * 128) <= 6 * ((C + 63) / 64 * 64) ? 128 : 64; }

inline int ilog2_exact(int v) {
    for (int l = 0; l < 31; ++l)
        if ((1 << l) == v) return l;
    return -1;
}
inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

}  // namespace

extern "C" {

// Sizes (in floats) of the packed forward / backward weight buffers for a layer.
int vf_conv_pack_sizes(int Cout, int Cin, int KS, long* fwd_floats, long* bwd_floats) {
    if (KS != 1 && KS != 3) return (int)hipErrorInvalidValue;
    const int ck = KS == 3 ? 8 : 32;
    *fwd_floats = (long)KS * KS * round_up(Cin, ck) * round_up(Cout, TCO);
    *bwd_floats = (long)KS * KS * round_up(Cout, ck) * round_up(Cin, TCO);
    return 0;
}

int vf_conv_pack_weights(const float* w_oihw, float* w_fwd, float* w_bwd, int Cout, int Cin, int KS,
                         void* stream) {
    if (KS != 1 && KS != 3) return (int)hipErrorInvalidValue;
    const int ck = KS == 3 ? 8 : 32, NT = KS * KS;
    const size_t nf = (size_t)NT * round_up(Cin, ck) * round_up(Cout, TCO);
    const size_t nb = w_bwd ? (size_t)NT * round_up(Cout, ck) * round_up(Cin, TCO) : 0;
    const size_t n = nf + nb;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, w_fwd, w_bwd, Cout, Cin, KS, nf, nb);
    VF_RETURN_LAST_ERROR();
}

// Pack every layer described by the device table `desc` ([nlayers][9] int64:
// {w_ptr, wf_ptr, wb_ptr, Cout, Cin, KS, nf, nb, first_block}) in one launch of total_blocks x 256.
int vf_conv_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream) {
    if (nlayers <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(pack_weights_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackDesc*)desc, nlayers);
    VF_RETURN_LAST_ERROR();
}

// y[S][Cout][H][W] = conv(x) (+bias +view_bias +residual).  H == W == power of two in [8,128]
// is the OUTPUT size.  `w_packed` comes from vf_conv_pack_weights.
// mode 0: x is [S][Cin][H][W]; 1: stride 2, x is [S][Cin][2H][2W]; 2: x is [S][Cin][H/2][W/2]
// nearest-upsampled on the fly; 4: x = dY [S][Cin][H][W] of a stride-2 conv, y = dX [S][Cout][2H][2W] (w_packed =
// the dgrad pack, no epilogue operands).
// ws / ws_floats: optional split-K workspace (see vf_conv_fwd_ws_floats); NULL disables split-K.
struct GnFuse {
    const float* gamma;
    const float* beta;
    float* out;
    float* stats;
    int groups, silu, store_y;
    float eps;
};

static int conv_fwd_impl(const float* x, const float* x2, int C1, const float* w_packed, const float* bias,
                         const float* view_bias, const float* residual, float* y, float* y2, int C1o, float* ws,
                         long ws_floats, int S, int Cin, int Cout, int H, int W, int KS, int mode, void* stream,
                         const GnFuse* gn = nullptr) {
    if (S <= 0) return 0;
    const int lw = ilog2_exact(W);
    if (H != W || lw < 3 || lw > 7 || (KS != 1 && KS != 3)) return (int)hipErrorInvalidValue;
    if (KS == 1 && mode != 0) return (int)hipErrorInvalidValue;
    if ((x2 || y2) && KS != 1) return (int)hipErrorInvalidValue;
    if (x2 && (C1 <= 0 || C1 >= Cin || C1 % 32 != 0)) return (int)hipErrorInvalidValue;     // chunk-aligned split
    if (y2 && (C1o <= 0 || C1o >= Cout || C1o % TCO != 0)) return (int)hipErrorInvalidValue; // tile-aligned split
    ConvArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.x2 = x2; a.C1 = C1; a.y2 = y2; a.C1o = C1o;
    if (y2) ws = nullptr;                                 // (the split-K reduce writes one destination)
    a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CinP = round_up(Cin, KS == 3 ? 8 : 32);
    a.CoutP = round_up(Cout, TCO);
    a.ksplit = 1;
    a.ws = ws;
    if (gn) {
        if (gn->groups <= 0 || Cout % gn->groups != 0 || y2 || mode == 4) return (int)hipErrorInvalidValue;
        a.gn_gamma = gn->gamma; a.gn_beta = gn->beta; a.gn_out = gn->out; a.gn_stats = gn->stats;
        a.gn_groups = gn->groups; a.gn_silu = gn->silu; a.gn_store_y = gn->store_y; a.gn_eps = gn->eps;
    }
    hipStream_t st = (hipStream_t)stream;
    if (KS == 1 && !gn) {          // training-size grids: the dedicated 1x1 kernel (conv1x1.hip)
        C11Args c;
        c.x = x; c.x2 = x2; c.w = w_packed; c.bias = bias; c.vbias = view_bias; c.res = residual; c.y = y; c.y2 = y2;
        c.S = S; c.Cin = Cin; c.Cout = Cout; c.C1 = C1; c.C1o = C1o; c.HW = H * W; c.hwsh = 2 * lw; c.npx = S * H * W;
        c.nct = a.CoutP / TCO;
        static const bool off = getenv("VF_CONV1X1_OLD") != nullptr;       // (tuning aid: the generic kernel)
        if (!off && vfi_conv1x1_supported(c)) return vfi_conv1x1_launch(c, st);
    }
#define VF_CASE(KS_, LW_, M_) \
    if (KS == KS_ && lw == LW_ && mode == M_) return launch_conv<KS_, LW_, M_>(a, st, ws_floats);
    VF_CASE(3, 3, 0) VF_CASE(3, 4, 0) VF_CASE(3, 5, 0) VF_CASE(3, 6, 0) VF_CASE(3, 7, 0)
    VF_CASE(3, 3, 1) VF_CASE(3, 4, 1) VF_CASE(3, 5, 1) VF_CASE(3, 6, 1)
    VF_CASE(3, 4, 2) VF_CASE(3, 5, 2) VF_CASE(3, 6, 2) VF_CASE(3, 7, 2)
    VF_CASE(3, 3, 4) VF_CASE(3, 4, 4) VF_CASE(3, 5, 4) VF_CASE(3, 6, 4)
    VF_CASE(1, 3, 0) VF_CASE(1, 4, 0) VF_CASE(1, 5, 0) VF_CASE(1, 6, 0) VF_CASE(1, 7, 0)
#undef VF_CASE
    return (int)hipErrorInvalidValue;
}

int vf_conv_fwd(const float* x, const float* w_packed, const float* bias, const float* view_bias,
                const float* residual, float* y, float* ws, long ws_floats, int S, int Cin, int Cout, int H,
                int W, int KS, int mode, void* stream) {
    return conv_fwd_impl(x, nullptr, Cin, w_packed, bias, view_bias, residual, y, nullptr, Cout, ws, ws_floats, S, Cin,
                         Cout, H, W, KS, mode, stream);
}

// Inference fusion: a = [Swish](GroupNorm(groups)(conv(x) + bias + view_bias + residual)) with the GroupNorm evaluated by
// the conv's split-K reduce launch where the conv runs split-K (small S: the sampler), by a plain GroupNorm launch
// otherwise.  y (always a valid [S][Cout][H][W] buffer) holds the conv output afterwards only if store_y != 0 or the
// unfused route was taken; gn_stats = scratch of 2*S*groups floats.
int vf_conv_fwd_gn(const float* x, const float* w_packed, const float* bias, const float* view_bias, const float* residual,
                   float* y, int store_y, const float* gn_gamma, const float* gn_beta, float* a_out, float* gn_stats,
                   int groups, float eps, int silu, float* ws, long ws_floats, int S, int Cin, int Cout, int H, int W,
                   int KS, int mode, void* stream) {
    GnFuse gn{gn_gamma, gn_beta, a_out, gn_stats, groups, silu, store_y, eps};
    return conv_fwd_impl(x, nullptr, Cin, w_packed, bias, view_bias, residual, y, nullptr, Cout, ws, ws_floats, S, Cin,
                         Cout, H, W, KS, mode, stream, &gn);
}

// 1x1 conv whose input is the never-materialised channel concatenation [x1 (C1 channels) | x2 (Cin - C1)]
// (decoder skip connections, reference unet.py:134 + 238): forward, dgrad (two destinations) and wgrad.
// C1 must be a multiple of 64.
int vf_conv1x1_cat_fwd(const float* x1, const float* x2, int C1, const float* w_packed, const float* bias, float* y,
                       float* ws, long ws_floats, int S, int Cin, int Cout, int H, int W, void* stream) {
    return conv_fwd_impl(x1, x2, C1, w_packed, bias, nullptr, nullptr, y, nullptr, Cout, ws, ws_floats, S, Cin, Cout, H, W,
                         1, 0, stream);
}

// dx1 [S][C1][H][W], dx2 [S][Cin-C1][H][W] from dy [S][Cout][H][W]; w_packed_bwd = the dgrad pack of the layer
int vf_conv1x1_cat_dgrad(const float* dy, const float* w_packed_bwd, float* dx1, float* dx2, int C1, int S, int Cin,
                         int Cout, int H, int W, void* stream) {
    return conv_fwd_impl(dy, nullptr, Cout, w_packed_bwd, nullptr, nullptr, nullptr, dx1, dx2, C1, nullptr, 0, S, Cout, Cin,
                         H, W, 1, 0, stream);
}

// Workspace floats vf_conv_fwd wants for its split-K path at this shape (0: no split-K, the
// natural grid already fills the chip).  Upper bound over the tile choices.
long vf_conv_fwd_ws_floats(int S, int Cin, int Cout, int H, int W, int KS) {
    const long tiles64 = ((long)S * H * W + 63) / 64;
    const long nblk_min = ((long)S * H * W + 127) / 128 * (round_up(Cout, TCO) / TCO);
    (void)tiles64;
    const int nch = round_up(Cin, KS == 3 ? 8 : 32) / (KS == 3 ? 8 : 32);
    const int ks = choose_ksplit((int)nblk_min, nch);
    return ks > 1 ? (long)ks * S * Cout * H * W : 0;
}

// Workspace floats needed by vf_conv_wgrad for the preferred split (a smaller workspace is
// accepted down to 1 slab and just reduces the split-K factor).
long vf_conv_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W, int KS) {
    if (KS == 1) {
        const int tm = wgrad1_tile(Cout), tn = wgrad1_tile(Cin);
        const long slab1 = (long)round_up(Cout, tm) * round_up(Cin, tn);
        long z1 = WGRAD_TARGET_WGS / ((round_up(Cout, tm) / tm) * (round_up(Cin, tn) / tn));
        const long nsteps = (long)S * (H * W / 64);
        if (z1 > nsteps / 4) z1 = nsteps / 4;
        if (z1 < 1) z1 = 1;
        return z1 * slab1;
    }
    const long slab = (long)KS * KS * round_up(Cout, TCO) * round_up(Cin, 32);
    const int nco = round_up(Cout, TCO) / TCO, nci = round_up(Cin, 32) / 32;
    long z = (WGRAD_TARGET_WGS + nco * nci - 1) / (nco * nci);
    long ntiles = ((long)S * H * W + 127) / 128;
    if (z > ntiles) z = ntiles;
    if (z < 1) z = 1;
    return z * slab;
}

// dw[Cout][Cin][KS][KS] = sum_{s,p} dy[s][co][p] * x_as_seen_by_the_conv[s][ci][p (+) tap]
// (H, W = OUTPUT size = dy size; mode as in vf_conv_fwd, 0..2).
static int conv_wgrad_impl(const float* x, const float* x2, int C1, const float* dy, float* dw, float* ws,
                           long ws_floats, int S, int Cin, int Cout, int H, int W, int KS, int mode, void* stream,
                           long long* desc = nullptr, int* nblocks = nullptr) {
    if (desc) *nblocks = 0;
    if (S <= 0) return 0;
    const int lw = ilog2_exact(W);
    if (H != W || lw < 3 || lw > 7 || (KS != 1 && KS != 3)) return (int)hipErrorInvalidValue;
    if (KS == 1 && mode != 0) return (int)hipErrorInvalidValue;
    if (x2 && (KS != 1 || C1 <= 0 || C1 >= Cin || C1 % 64 != 0)) return (int)hipErrorInvalidValue;
    if (KS == 1) {                                        // plain GEMM: the large-tile kernel
        Wgrad1Args g;
        g.x = x; g.x2 = x2; g.dy = dy; g.ws = ws; g.S = S; g.Cin = Cin; g.Cout = Cout; g.C1 = C1; g.HW = H * W;
        const int tm = wgrad1_tile(Cout), tn = wgrad1_tile(Cin);
        g.CoutP = round_up(Cout, tm); g.CinQ = round_up(Cin, tn);
        g.nsteps = S * (H * W / 64);
        const int nco = g.CoutP / tm, nci = g.CinQ / tn;
        const size_t slab = (size_t)g.CoutP * g.CinQ;
        int z = WGRAD_TARGET_WGS / (nco * nci);
        if (z > g.nsteps / 4) z = g.nsteps / 4;           // >= 4 steps per slice: slab traffic vs fill on 8x8 maps
        if (z < 1) z = 1;
        if ((size_t)ws_floats < slab) return (int)hipErrorInvalidValue;
        if ((size_t)z > (size_t)ws_floats / slab) z = (int)((size_t)ws_floats / slab);
        g.steps_per_slice = (g.nsteps + z - 1) / z;
        z = (g.nsteps + g.steps_per_slice - 1) / g.steps_per_slice;
        hipStream_t st1 = (hipStream_t)stream;
        const dim3 grid(nco, nci, z);
        if (tm == 128 && tn == 128) hipLaunchKernelGGL((conv1x1_wgrad_kernel<128, 128>), grid, dim3(256), 0, st1, g);
        else if (tm == 128) hipLaunchKernelGGL((conv1x1_wgrad_kernel<128, 64>), grid, dim3(256), 0, st1, g);
        else if (tn == 128) hipLaunchKernelGGL((conv1x1_wgrad_kernel<64, 128>), grid, dim3(256), 0, st1, g);
        else hipLaunchKernelGGL((conv1x1_wgrad_kernel<64, 64>), grid, dim3(256), 0, st1, g);
        const int total = Cout * Cin;
        if (desc) {
            *nblocks = wgrad_reduce_row(desc, ws, dw, z, 1, Cout, Cin, g.CoutP, g.CinQ);
            VF_RETURN_LAST_ERROR();
        }
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, st1, ws, dw, z, 1, Cout, Cin,
                           g.CoutP, g.CinQ);
        VF_RETURN_LAST_ERROR();
    }
    WgradArgs a;
    a.x = x; a.x2 = x2; a.C1 = C1; a.dy = dy; a.ws = ws; a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CoutP = round_up(Cout, TCO);
    a.CinQ = round_up(Cin, 32);
    hipStream_t st = (hipStream_t)stream;
#define VF_CASE(KS_, LW_, M_) \
    if (KS == KS_ && lw == LW_ && mode == M_) return launch_wgrad<KS_, LW_, M_>(a, dw, (size_t)ws_floats, st, desc, nblocks);
    VF_CASE(3, 3, 0) VF_CASE(3, 4, 0) VF_CASE(3, 5, 0) VF_CASE(3, 6, 0) VF_CASE(3, 7, 0)
    VF_CASE(3, 3, 1) VF_CASE(3, 4, 1) VF_CASE(3, 5, 1) VF_CASE(3, 6, 1)
    VF_CASE(3, 4, 2) VF_CASE(3, 5, 2) VF_CASE(3, 6, 2) VF_CASE(3, 7, 2)
#undef VF_CASE
    return (int)hipErrorInvalidValue;
}

int vf_conv_wgrad(const float* x, const float* dy, float* dw, float* ws, long ws_floats, int S, int Cin,
                  int Cout, int H, int W, int KS, int mode, void* stream) {
    return conv_wgrad_impl(x, nullptr, Cin, dy, dw, ws, ws_floats, S, Cin, Cout, H, W, KS, mode, stream);
}

int vf_conv1x1_cat_wgrad(const float* x1, const float* x2, int C1, const float* dy, float* dw, float* ws,
                         long ws_floats, int S, int Cin, int Cout, int H, int W, void* stream) {
    return conv_wgrad_impl(x1, x2, C1, dy, dw, ws, ws_floats, S, Cin, Cout, H, W, 1, 0, stream);
}

// vf_conv_wgrad / vf_conv1x1_cat_wgrad without their follow-up launch: the main kernel only; desc9 (HOST memory, 9 x int64)
// receives the row that vf_wino44_reduce_multi needs for this layer, *nblocks its workgroup count.  ws must stay untouched
// until that launch.
int vf_conv_wgrad_main(const float* x, const float* dy, float* dw, float* ws, long ws_floats, int S, int Cin, int Cout,
                       int H, int W, int KS, int mode, long long* desc9, int* nblocks, void* stream) {
    if (!desc9 || !nblocks) return (int)hipErrorInvalidValue;
    return conv_wgrad_impl(x, nullptr, Cin, dy, dw, ws, ws_floats, S, Cin, Cout, H, W, KS, mode, stream, desc9, nblocks);
}

int vf_conv1x1_cat_wgrad_main(const float* x1, const float* x2, int C1, const float* dy, float* dw, float* ws,
                              long ws_floats, int S, int Cin, int Cout, int H, int W, long long* desc9, int* nblocks,
                              void* stream) {
    if (!desc9 || !nblocks) return (int)hipErrorInvalidValue;
    return conv_wgrad_impl(x1, x2, C1, dy, dw, ws, ws_floats, S, Cin, Cout, H, W, 1, 0, stream, desc9, nblocks);
}

int vf_sumpool2(const float* x, float* y, long n_out, int Wo, void* stream) {
    if (n_out <= 0) return 0;
    hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, (size_t)n_out, Wo);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
