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
//     3: zero-dilated input (dgrad of the stride-2 conv)
// * Epilogue fuses bias[co] + per-view bias[s][co] (time/angle embedding) + residual.
// * wgrad: D[co][ci] per tap = sum_pixels dY[co][p] * X[ci][p (+) tap]; split-K over pixel
//   tiles into slabs, reduced by a second deterministic kernel (no float atomics).
//
// Bound: fp32 matrix rate (157.3 TF), not HBM: AI of these layers is 144-960 flop/B.
#include "common.h"

namespace {

constexpr int TPIX = 128;  // output pixels per workgroup tile
constexpr int TCO = 64;    // output channels per workgroup tile
constexpr int WGRAD_TARGET_WGS = 512;  // split-K slices are sized for ~2 workgroups per CU

template <int KS, int LOGW, int MODE>
struct Geo {
    static constexpr int W = 1 << LOGW;
    static constexpr int H = W;
    static constexpr int HW = W * H;
    static constexpr int IM = HW >= TPIX ? 1 : TPIX / HW;        // images per tile
    static constexpr int TH = HW >= TPIX ? TPIX / W : H;         // output rows per image per tile
    static constexpr int TPI = HW >= TPIX ? HW / TPIX : 1;       // tiles per image
    static constexpr int PAD = KS / 2;
    static constexpr int PH = MODE == 1 ? 2 * TH + 1 : TH + 2 * PAD;
    static constexpr int PW = KS == 1 ? W : (MODE == 1 ? 2 * W + 4 : W + 8);
    static constexpr int CO = KS == 1 ? 0 : 3;                   // LDS column of patch x = -1
    static constexpr int Q = (MODE == 1 ? 2 * W : W) / 4;        // interior float4 per patch row
    static constexpr int PS = IM * PH * PW;                      // floats per channel plane
    static constexpr int SH = MODE == 1 ? 2 * H : ((MODE == 2 || MODE == 3) ? H / 2 : H);  // source
    static constexpr int SW = MODE == 1 ? 2 * W : ((MODE == 2 || MODE == 3) ? W / 2 : W);
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
    const float* w;      // packed [tap][CinP][CoutP]
    const float* bias;   // [Cout] or null
    const float* vbias;  // [S][Cout] or null
    const float* res;    // [S][Cout][H][W] or null
    float* y;
    int S, Cin, Cout, CinP, CoutP;
};

// Global -> register load of one float4 of the input patch (zero outside the image / tensor).
template <int KS, int LOGW, int MODE>
__device__ __forceinline__ float4 load_patch4(const float* __restrict__ x, int S, int Cin, int s, int ci,
                                              int r0, int pr, int q) {
    using G = Geo<KS, LOGW, MODE>;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= S || ci >= Cin) return v;
    const size_t plane = ((size_t)s * Cin + ci) * (size_t)(G::SH * G::SW);
    if (MODE == 0) {
        const int gy = r0 + pr - G::PAD;
        if (gy >= 0 && gy < G::H) v = *reinterpret_cast<const float4*>(x + plane + gy * G::SW + 4 * q);
    } else if (MODE == 1) {
        const int gy = 2 * r0 + pr - 1;
        if (gy >= 0 && gy < G::SH) v = *reinterpret_cast<const float4*>(x + plane + gy * G::SW + 4 * q);
    } else {
        const int uy = r0 + pr - G::PAD;
        if (uy >= 0 && uy < G::H && (MODE == 2 || (uy & 1) == 0)) {
            const float2 t = *reinterpret_cast<const float2*>(x + plane + (uy >> 1) * G::SW + 2 * q);
            v = MODE == 2 ? make_float4(t.x, t.x, t.y, t.y) : make_float4(t.x, 0.f, t.y, 0.f);
        }
    }
    return v;
}

template <int KS, int LOGW, int MODE>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    using G = Geo<KS, LOGW, MODE>;
    constexpr int CK = KS == 3 ? 8 : 32;
    constexpr int NT = KS * KS;
    constexpr int NW4 = NT * CK * (TCO / 4);
    constexpr int NWL = (NW4 + 255) / 256;
    constexpr int NX4 = CK * G::IM * G::PH * G::Q;
    constexpr int NXL = (NX4 + 255) / 256;

    __shared__ __attribute__((aligned(16))) float wl[NT * CK * TCO];
    __shared__ __attribute__((aligned(16))) float xl[CK * G::PS];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, pw = wid >> 1;
    const int li = lane & 31, lh = lane >> 5;

    const int ncot = a.CoutP / TCO;
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int cot = logical % ncot;
    const int tile = logical / ncot;
    const int co0 = cot * TCO;
    const int s0 = G::IM > 1 ? tile * G::IM : tile / G::TPI;
    const int r0 = G::IM > 1 ? 0 : (tile % G::TPI) * G::TH;

    // zero the patch once: the left/right halo columns are never written again
    for (int i = tid; i < CK * G::PS; i += 256) xl[i] = 0.f;

    // Register staging: full 256-wide passes are unconditional and a ragged tail lives in its
    // own scalar so that the arrays are only ever indexed statically (stay in VGPRs).
    constexpr int NWF = NW4 / 256, NXF = NX4 / 256;
    constexpr bool WT = (NW4 % 256) != 0, XT = (NX4 % 256) != 0;
    float4 wreg[NWF > 0 ? NWF : 1], xreg[NXF > 0 ? NXF : 1];
    float4 wtail = make_float4(0.f, 0.f, 0.f, 0.f), xtail = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_w = [&](int e, int c0) -> float4 {
        const int q4 = e & 15, row = e >> 4;                     // row = tap*CK + ci
        const int tap = row / CK, ci = row - tap * CK;
        return *reinterpret_cast<const float4*>(a.w + ((size_t)tap * a.CinP + c0 + ci) * a.CoutP + co0 + 4 * q4);
    };
    auto load_x = [&](int e, int c0) -> float4 {
        const int q = e % G::Q;
        const int t1 = e / G::Q;
        const int pr = t1 % G::PH;
        const int t2 = t1 / G::PH;
        const int im = t2 % G::IM, ci = t2 / G::IM;
        return load_patch4<KS, LOGW, MODE>(a.x, a.S, a.Cin, s0 + im, c0 + ci, r0, pr, q);
    };
    auto store_x = [&](int e, const float4& v) {
        const int q = e % G::Q;
        const int t1 = e / G::Q;                                  // (ci*IM + im)*PH + pr
        *reinterpret_cast<float4*>(xl + t1 * G::PW + 4 * q + (KS == 1 ? 0 : 4)) = v;
    };
#define VF_LOAD_CHUNK(C0)                                                             \
    {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NWF; ++i) wreg[i] = load_w(tid + i * 256, (C0)); \
        if (WT && tid + NWF * 256 < NW4) wtail = load_w(tid + NWF * 256, (C0));        \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) xreg[i] = load_x(tid + i * 256, (C0)); \
        if (XT && tid + NXF * 256 < NX4) xtail = load_x(tid + NXF * 256, (C0));        \
    }
#define VF_STORE_CHUNK()                                                              \
    {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NWF; ++i)                                \
            *reinterpret_cast<float4*>(wl + 4 * (tid + i * 256)) = wreg[i];            \
        if (WT && tid + NWF * 256 < NW4) *reinterpret_cast<float4*>(wl + 4 * (tid + NWF * 256)) = wtail; \
        _Pragma("unroll") for (int i = 0; i < NXF; ++i) store_x(tid + i * 256, xreg[i]); \
        if (XT && tid + NXF * 256 < NX4) store_x(tid + NXF * 256, xtail);              \
    }

    f32x16 acc0 = {0}, acc1 = {0};
    const float* wb = wl + lh * TCO + cw * 32 + li;
    const int p0 = pw * 64 + li, p1 = p0 + 32;
    const float* xb0 = xl + lh * G::PS + G::pix_off(p0);
    const float* xb1 = xl + lh * G::PS + G::pix_off(p1);

    VF_LOAD_CHUNK(0);
    for (int c0 = 0; c0 < a.CinP; c0 += CK) {
        __syncthreads();                 // previous chunk's LDS reads (and the zero fill) done
        VF_STORE_CHUNK();
        __syncthreads();
        if (c0 + CK < a.CinP) VF_LOAD_CHUNK(c0 + CK);
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
            const int toff = (tap / KS) * G::PW + (tap % KS);
#pragma unroll
            for (int s = 0; s < CK / 2; ++s) {
                const float av = wb[(tap * CK + 2 * s) * TCO];
                const float b0 = xb0[2 * s * G::PS + toff];
                const float b1 = xb1[2 * s * G::PS + toff];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
            }
        }
    }

    // epilogue: lane = pixel (coalesced), register = output channel
#undef VF_LOAD_CHUNK
#undef VF_STORE_CHUNK
    auto epilogue = [&](const f32x16& acc, int nt) {
        const int p = pw * 64 + nt * 32 + li;
        const int s = G::IM > 1 ? s0 + (p >> (2 * LOGW)) : s0;
        const int pix = G::IM > 1 ? (p & (G::HW - 1)) : r0 * G::W + p;
        if (s < a.S) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (co < a.Cout) {
                    const size_t o = ((size_t)s * a.Cout + co) * G::HW + pix;
                    float v = acc[r];
                    if (a.bias) v += a.bias[co];
                    if (a.vbias) v += a.vbias[(size_t)s * a.Cout + co];
                    if (a.res) v += a.res[o];
                    a.y[o] = v;
                }
            }
        }
    };
    epilogue(acc0, 0);
    epilogue(acc1, 1);
}

// ----------------------------------------------------------------------------------------------
// weight gradient
struct WgradArgs {
    const float* x;
    const float* dy;
    float* ws;       // [slab][tap][CoutP][CinQ]
    int S, Cin, Cout, CoutP, CinQ;   // CoutP multiple of 64, CinQ multiple of 32
    int ntiles, tiles_per_slice;
};

template <int KS, int LOGW, int MODE>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
    using G = Geo<KS, LOGW, MODE>;
    constexpr int NT = KS * KS;
    constexpr int TCI = 32;
    constexpr int DYS = TPIX + 1;                 // odd row stride: conflict-free column reads
    constexpr int PSO = G::PS | 1;                // odd plane stride
    constexpr int NX4 = TCI * G::IM * G::PH * G::Q;
    constexpr int ND4 = TCO * (TPIX / 4);
    constexpr int OHW = G::HW;

    constexpr int LDSF = TCO * DYS + TCI * PSO;
    __shared__ float lds[LDSF];                   // one array: dY tile | input patch | k-half reduce
    float* const dyl = lds;
    float* const xl = lds + TCO * DYS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, kq = wid >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const int co0 = blockIdx.x * TCO, ci0 = blockIdx.y * TCI;
    const int t_begin = blockIdx.z * a.tiles_per_slice;
    const int t_end = min(a.ntiles, t_begin + a.tiles_per_slice);

    for (int i = tid; i < TCI * PSO; i += 256) xl[i] = 0.f;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x16){0};

    const float* ab = dyl + (cw * 32 + li) * DYS + kq * 64 + lh;
    const float* bb = xl + li * PSO + G::pix_off(kq * 64 + lh);

    for (int tile = t_begin; tile < t_end; ++tile) {
        const int s0 = G::IM > 1 ? tile * G::IM : tile / G::TPI;
        const int r0 = G::IM > 1 ? 0 : (tile % G::TPI) * G::TH;
        __syncthreads();
        // dY tile: 64 channels x 128 pixels
        for (int e = tid; e < ND4; e += 256) {
            const int q = e & 31, co = e >> 5;               // 32 float4 per row
            const int p = 4 * q;
            const int s = G::IM > 1 ? s0 + (p >> (2 * LOGW)) : s0;
            const int pix = G::IM > 1 ? (p & (OHW - 1)) : r0 * G::W + p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s < a.S && co0 + co < a.Cout)
                v = *reinterpret_cast<const float4*>(a.dy + ((size_t)s * a.Cout + co0 + co) * OHW + pix);
            float* d = dyl + co * DYS + p;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        // input patch: 32 channels
        for (int e = tid; e < NX4; e += 256) {
            const int q = e % G::Q;
            const int t1 = e / G::Q;
            const int pr = t1 % G::PH;
            const int t2 = t1 / G::PH;
            const int im = t2 % G::IM, ci = t2 / G::IM;
            const float4 v = load_patch4<KS, LOGW, MODE>(a.x, a.S, a.Cin, s0 + im, ci0 + ci, r0, pr, q);
            float* d = xl + ci * PSO + (im * G::PH + pr) * G::PW + 4 * q + (KS == 1 ? 0 : 4);
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const float av = ab[2 * s];
            const int d = G::pix_off(2 * s) - G::pix_off(0);
#pragma unroll
            for (int tap = 0; tap < NT; ++tap) {
                const float bv = bb[d + (tap / KS) * G::PW + (tap % KS)];
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
            }
        }
    }

    // the two pixel halves (kq) of each co-half are summed through LDS, in passes of NTP taps
    constexpr int NTP = NT > 1 ? (NT + 1) / 2 : 1;
    static_assert(2 * NTP * 16 * 64 <= LDSF, "k-half reduce buffer does not fit");
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += NTP) {
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
    const int slab = blockIdx.z;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            a.ws[(((size_t)slab * NT + tap) * a.CoutP + co) * a.CinQ + ci0 + li] = acc[tap][r];
        }
    }
}

// dw[co][ci][tap] = sum_slab ws[slab][tap][co][ci]
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nslab, int NT,
                                    int Cout, int Cin, int CoutP, int CinQ) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over (tap, co, ci), ci fastest
    const int total = NT * Cout * Cin;
    if (idx >= total) return;
    const int ci = idx % Cin;
    const int t = idx / Cin;
    const int co = t % Cout, tap = t / Cout;
    const size_t stride = (size_t)NT * CoutP * CinQ;
    const float* p = ws + ((size_t)tap * CoutP + co) * CinQ + ci;
    float acc = 0.f;
    for (int s = 0; s < nslab; ++s) acc += p[s * stride];
    dw[((size_t)co * Cin + ci) * NT + tap] = acc;
}

// OIHW -> packed forward [tap][CinP][CoutP] and backward (dgrad) [tap'][CoutPk][CinPm] with
// flipped taps; zero padded.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb,
                                    int Cout, int Cin, int NT, int CinPk, int CoutPm, int CoutPk, int CinPm) {
    const size_t nf = (size_t)NT * CinPk * CoutPm, nb = (size_t)NT * CoutPk * CinPm;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < nf) {
        const int co = idx % CoutPm;
        const size_t t = idx / CoutPm;
        const int ci = t % CinPk, tap = t / CinPk;
        wf[idx] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * NT + tap] : 0.f;
    } else if (idx < nf + nb && wb) {
        const size_t j = idx - nf;
        const int ci = j % CinPm;
        const size_t t = j / CinPm;
        const int co = t % CoutPk, tap = t / CoutPk;
        wb[j] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * NT + (NT - 1 - tap)] : 0.f;
    }
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

template <int KS, int LOGW, int MODE>
int launch_conv(const ConvArgs& a, hipStream_t st) {
    using G = Geo<KS, LOGW, MODE>;
    const int ntiles = G::IM > 1 ? (a.S + G::IM - 1) / G::IM : a.S * G::TPI;
    const int nblk = ntiles * (a.CoutP / TCO);
    hipLaunchKernelGGL((conv_mfma_kernel<KS, LOGW, MODE>), dim3(nblk), dim3(256), 0, st, a);
    VF_RETURN_LAST_ERROR();
}

template <int KS, int LOGW, int MODE>
int launch_wgrad(WgradArgs a, float* dw, size_t ws_floats, hipStream_t st) {
    using G = Geo<KS, LOGW, MODE>;
    constexpr int NT = KS * KS;
    a.ntiles = G::IM > 1 ? (a.S + G::IM - 1) / G::IM : a.S * G::TPI;
    const int nco = a.CoutP / TCO, nci = a.CinQ / 32;
    const size_t slab_floats = (size_t)NT * a.CoutP * a.CinQ;
    int z = (WGRAD_TARGET_WGS + nco * nci - 1) / (nco * nci);
    if (z > a.ntiles) z = a.ntiles;
    const size_t zmax = ws_floats / slab_floats;
    if (zmax < 1) return (int)hipErrorInvalidValue;
    if ((size_t)z > zmax) z = (int)zmax;
    a.tiles_per_slice = (a.ntiles + z - 1) / z;
    z = (a.ntiles + a.tiles_per_slice - 1) / a.tiles_per_slice;
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, LOGW, MODE>), dim3(nco, nci, z), dim3(256), 0, st, a);
    const int total = NT * a.Cout * a.Cin;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, a.ws, dw, z, NT,
                       a.Cout, a.Cin, a.CoutP, a.CinQ);
    VF_RETURN_LAST_ERROR();
}

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
    const int CinPk = round_up(Cin, ck), CoutPm = round_up(Cout, TCO);
    const int CoutPk = round_up(Cout, ck), CinPm = round_up(Cin, TCO);
    const size_t n = (size_t)NT * CinPk * CoutPm + (w_bwd ? (size_t)NT * CoutPk * CinPm : 0);
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, w_fwd, w_bwd, Cout, Cin, NT, CinPk, CoutPm, CoutPk, CinPm);
    VF_RETURN_LAST_ERROR();
}

// y[S][Cout][H][W] = conv(x) (+bias +view_bias +residual).  H == W == power of two in [8,128]
// is the OUTPUT size.  `w_packed` is [tap][round_up(Cin,ck)][round_up(Cout,64)].
// mode 0: x is [S][Cin][H][W]; 1: stride 2, x is [S][Cin][2H][2W]; 2: x is [S][Cin][H/2][W/2]
// nearest-upsampled on the fly; 3: x is [S][Cin][H/2][W/2] zero-dilated on the fly.
int vf_conv_fwd(const float* x, const float* w_packed, const float* bias, const float* view_bias,
                const float* residual, float* y, int S, int Cin, int Cout, int H, int W, int KS, int mode,
                void* stream) {
    if (S <= 0) return 0;
    const int lw = ilog2_exact(W);
    if (H != W || lw < 3 || lw > 7 || (KS != 1 && KS != 3)) return (int)hipErrorInvalidValue;
    if (KS == 1 && mode != 0) return (int)hipErrorInvalidValue;
    ConvArgs a;
    a.x = x; a.w = w_packed; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CinP = round_up(Cin, KS == 3 ? 8 : 32);
    a.CoutP = round_up(Cout, TCO);
    hipStream_t st = (hipStream_t)stream;
#define VF_CASE(KS_, LW_, M_) \
    if (KS == KS_ && lw == LW_ && mode == M_) return launch_conv<KS_, LW_, M_>(a, st);
    VF_CASE(3, 3, 0) VF_CASE(3, 4, 0) VF_CASE(3, 5, 0) VF_CASE(3, 6, 0) VF_CASE(3, 7, 0)
    VF_CASE(3, 3, 1) VF_CASE(3, 4, 1) VF_CASE(3, 5, 1) VF_CASE(3, 6, 1)
    VF_CASE(3, 4, 2) VF_CASE(3, 5, 2) VF_CASE(3, 6, 2) VF_CASE(3, 7, 2)
    VF_CASE(3, 4, 3) VF_CASE(3, 5, 3) VF_CASE(3, 6, 3) VF_CASE(3, 7, 3)
    VF_CASE(1, 3, 0) VF_CASE(1, 4, 0) VF_CASE(1, 5, 0) VF_CASE(1, 6, 0) VF_CASE(1, 7, 0)
#undef VF_CASE
    return (int)hipErrorInvalidValue;
}

// Workspace floats needed by vf_conv_wgrad for the preferred split (a smaller workspace is
// accepted down to 1 slab and just reduces the split-K factor).
long vf_conv_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W, int KS) {
    const long slab = (long)KS * KS * round_up(Cout, TCO) * round_up(Cin, 32);
    const int nco = round_up(Cout, TCO) / TCO, nci = round_up(Cin, 32) / 32;
    long z = (WGRAD_TARGET_WGS + nco * nci - 1) / (nco * nci);
    long ntiles = ((long)S * H * W + TPIX - 1) / TPIX;
    if (z > ntiles) z = ntiles;
    if (z < 1) z = 1;
    return z * slab;
}

// dw[Cout][Cin][KS][KS] = sum_{s,p} dy[s][co][p] * x_as_seen_by_the_conv[s][ci][p (+) tap]
// (H, W = OUTPUT size = dy size; mode as in vf_conv_fwd, 0..2).
int vf_conv_wgrad(const float* x, const float* dy, float* dw, float* ws, long ws_floats, int S, int Cin,
                  int Cout, int H, int W, int KS, int mode, void* stream) {
    if (S <= 0) return 0;
    const int lw = ilog2_exact(W);
    if (H != W || lw < 3 || lw > 7 || (KS != 1 && KS != 3)) return (int)hipErrorInvalidValue;
    if (KS == 1 && mode != 0) return (int)hipErrorInvalidValue;
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = ws; a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CoutP = round_up(Cout, TCO);
    a.CinQ = round_up(Cin, 32);
    hipStream_t st = (hipStream_t)stream;
#define VF_CASE(KS_, LW_, M_) \
    if (KS == KS_ && lw == LW_ && mode == M_) return launch_wgrad<KS_, LW_, M_>(a, dw, (size_t)ws_floats, st);
    VF_CASE(3, 3, 0) VF_CASE(3, 4, 0) VF_CASE(3, 5, 0) VF_CASE(3, 6, 0) VF_CASE(3, 7, 0)
    VF_CASE(3, 3, 1) VF_CASE(3, 4, 1) VF_CASE(3, 5, 1) VF_CASE(3, 6, 1)
    VF_CASE(3, 4, 2) VF_CASE(3, 5, 2) VF_CASE(3, 6, 2) VF_CASE(3, 7, 2)
    VF_CASE(1, 3, 0) VF_CASE(1, 4, 0) VF_CASE(1, 5, 0) VF_CASE(1, 6, 0) VF_CASE(1, 7, 0)
#undef VF_CASE
    return (int)hipErrorInvalidValue;
}

int vf_sumpool2(const float* x, float* y, long n_out, int Wo, void* stream) {
    if (n_out <= 0) return 0;
    hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       x, y, (size_t)n_out, Wo);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
