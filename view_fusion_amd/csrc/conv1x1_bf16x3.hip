// EXPERIMENT, default OFF (VF_BF16X3=1 on the host side selects it): the 1x1 convolutions (reference model/unet.py:238,
// 255, 256 -- residual, qkv and attention-output projections) with fp32-ACCURATE products on the bf16 matrix path.
//
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)            (3 x 8 significant bits)
//   a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)    six v_mfma_f32_32x32x16_bf16, fp32 accumulation,
//                                                                smallest terms first
// One 32x32x16 tile against fp64 (tools/mfma_bf16x3.hip): 8 fp32 MFMAs 4.3e-7, these 6 bf16 MFMAs 2.5e-7 -- the
// dropped terms (a2 b3, a3 b2, a3 b3) are below 2^-24 of the product.  The matrix pipe runs bf16 at 16x the fp32-input
// rate, so six products per K=16 block cost 6/16 of eight fp32 MFMAs: 2.67x the fp32 MFMA peak in fp32-equivalent FLOPs.
// The arithmetic type is still fp32 in and out; a line measured with this kernel must say "bf16x3" in its dtype.
//
// GEMM  Y[co][n] = sum_ci W[co][ci] X[ci][n],  n = (view, pixel) flattened, per workgroup 128 co x 128 n, 4 waves =
// 2 (co) x 2 (n), each 64 x 64 = four 32x32 accumulators.  K in chunks of 32 ci (two MFMA k-steps):
//   * weights arrive pre-split and packed (vf_conv1x1_bf16x3_pack): [co tile][chunk][k-step 2][row block 4][plane 3]
//     [lane 64][8 bf16] -- a wave's A fragment is one contiguous 1 KB line, loaded straight into registers;
//   * activations: thread (n, k group of 8 ci) loads its eight fp32 values (coalesced over n), splits them into three
//     bf16x8 and writes three 16-byte LDS rows [plane][k group][n][8 bf16] = exactly the MFMA B fragment; done once per
//     workgroup, double buffered, one barrier per chunk;
//   * epilogue: bias + per-view bias + residual, fp32 stores (32 consecutive n per lane row).
// Also serves the dgrad (weights packed transposed) and the never-materialised decoder concatenation (x | x2).
#include "common.h"

namespace {
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

inline int rup3(int v, int m) { return (v + m - 1) / m * m; }

constexpr int B3_TCO = 128, B3_TN = 128, B3_KC = 32;

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {          // two fp32 -> two bf16 (RNE), one instruction
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }

// eight fp32 -> three bf16x8 (as 4 packed dwords each)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& p1, u32x4& p2, u32x4& p3) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float a = x[2 * i], b = x[2 * i + 1];
        p1[i] = pk_bf16(a, b);
        const float ra = a - bf_lo(p1[i]), rb = b - bf_hi(p1[i]);
        p2[i] = pk_bf16(ra, rb);
        p3[i] = pk_bf16(ra - bf_lo(p2[i]), rb - bf_hi(p2[i]));
    }
}

struct B3Args {
    const float* x;
    const float* x2;          // second input tensor of a channel concatenation, or null
    const unsigned* w3;       // packed split weights
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    float* y2;                // second output tensor (dgrad of a concatenation), or null
    int S, Cin, Cout, C1in, C1out, HW, hwsh, N, nchunks;
};

__global__ __launch_bounds__(256, 2) void conv1x1_bf16x3_kernel(B3Args a) {
    constexpr int PSZ = B3_KC / 8 * B3_TN * 4;                 // dwords of one plane of one buffer: [kg 4][n 128][4 dwords]
    __shared__ __attribute__((aligned(16))) unsigned Bl[2 * 3 * PSZ];       // 2 x 3 x 8 KB = 48 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 1, wn = wid >> 1;                     // wave position: co half, n half
    const int li = lane & 31, lh = lane >> 5;
    const unsigned nblk_co = (a.Cout + B3_TCO - 1) / B3_TCO;
    const unsigned lgc = xcd_remap(blockIdx.x, gridDim.x);      // consecutive logical ids (the co tiles of one n tile) share an XCD
    const int co0 = (lgc % nblk_co) * B3_TCO, n0 = (lgc / nblk_co) * B3_TN;

    // staging duty: item e = tid + 256 i (i < 2) -> (k group kg = e >> 7, column n = e & 127); kg is wave-uniform, so the
    // channel index -- and with it the source tensor of a concatenation and the bounds test -- is scalar
    const int sn = tid & 127, skg0 = wid >> 1;
    const int gn = n0 + sn;
    const int ss = min(gn, a.N - 1) >> a.hwsh, sp = min(gn, a.N - 1) & (a.HW - 1);
    const int cin_in1 = a.x2 ? a.C1in : a.Cin;                 // channels held by the first input tensor
    const float* const px1 = a.x + (size_t)ss * cin_in1 * a.HW + sp;
    const float* const px2 = a.x2 ? a.x2 + (size_t)ss * (a.Cin - a.C1in) * a.HW + sp : nullptr;
    float xv[2][8];
    auto load_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ci0 = c * B3_KC + 8 * (skg0 + 2 * i);   // scalar
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int ci = ci0 + j;
                if (ci < a.Cin) {                              // scalar branch
                    const float* q = ci < cin_in1 ? px1 + (size_t)ci * a.HW : px2 + (size_t)(ci - cin_in1) * a.HW;
                    xv[i][j] = *q;                             // (column beyond N: clamped address, its result is never stored)
                } else {
                    xv[i][j] = 0.f;
                }
            }
        }
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            u32x4 p1, p2, p3;
            split8(xv[i], p1, p2, p3);
            unsigned* d = Bl + buf * 3 * PSZ + ((skg0 + 2 * i) * B3_TN + sn) * 4;
            *reinterpret_cast<u32x4*>(d) = p1;
            *reinterpret_cast<u32x4*>(d + PSZ) = p2;
            *reinterpret_cast<u32x4*>(d + 2 * PSZ) = p3;
        }
    };

    // A fragments of (chunk c, k-step ks, row block rb, plane pl): 64 lanes x 16 bytes, contiguous
    const unsigned* wbase = a.w3 + ((size_t)(co0 / B3_TCO) * a.nchunks) * (2 * 4 * 3 * 64 * 4) + lane * 4;
    auto load_a = [&](int c, int ks, int rbi, u32x4 (&f)[3]) {
        const unsigned* p = wbase + (((size_t)c * 2 + ks) * 4 + (2 * wm + rbi)) * (3 * 64 * 4);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const u32x4*>(p + pl * 64 * 4);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x16){0};

    // (A software-pipelined variant -- fragments of k-step t+1 requested before the MFMAs of k-step t -- was measured:
    // 206 registers, two workgroups per CU instead of three, 10-25 % SLOWER on every shape.  Occupancy hides the
    // fragment latency better here.)
    load_b(0);
    store_b(0);
    if (a.nchunks > 1) load_b(1);
    __syncthreads();
    for (int c = 0; c < a.nchunks; ++c) {
        const int buf = c & 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[2][3], bfr[2][3];
#pragma unroll
            for (int rbi = 0; rbi < 2; ++rbi) load_a(c, ks, rbi, af[rbi]);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    bfr[cb][pl] = *reinterpret_cast<const u32x4*>(
                        Bl + buf * 3 * PSZ + pl * PSZ + ((2 * ks + lh) * B3_TN + wn * 64 + cb * 32 + li) * 4);
            if (ks == 0 && c + 1 < a.nchunks) store_b(buf ^ 1);         // chunk c+1 -> the other buffer
            if (ks == 1 && c + 2 < a.nchunks) load_b(c + 2);
#pragma unroll
            for (int rbi = 0; rbi < 2; ++rbi)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[rbi][0]), a2 = __builtin_bit_cast(bf16x8, af[rbi][1]),
                                 a3 = __builtin_bit_cast(bf16x8, af[rbi][2]);
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, bfr[cb][0]), b2 = __builtin_bit_cast(bf16x8, bfr[cb][1]),
                                 b3 = __builtin_bit_cast(bf16x8, bfr[cb][2]);
                    f32x16 d = acc[rbi][cb];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d, 0, 0, 0);
                    acc[rbi][cb] = d;
                }
        }
        __syncthreads();
    }

    // epilogue: acc[rbi][cb][r] = Y[co0 + 64 wm + 32 rbi + (r&3) + 8 (r>>2) + 4 lh][n0 + 64 wn + 32 cb + li]
    // (a 64-channel wave range lies in ONE output tensor: the split point of a concatenation is a multiple of 64)
    const int cout_in1 = a.y2 ? a.C1out : a.Cout;
    const int cw0 = co0 + wm * 64;
    const bool second = a.y2 && cw0 >= a.C1out;
    float* const ybase = second ? a.y2 : a.y;
    const int cs = second ? a.Cout - a.C1out : cout_in1, cofs = second ? a.C1out : 0;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int n = n0 + wn * 64 + cb * 32 + li;
        if (n >= a.N) continue;
        const int s = n >> a.hwsh, p = n & (a.HW - 1);
        const size_t ob = ((size_t)s * cs + (cw0 - cofs + 4 * lh)) * a.HW + p;
        const float* vb = a.vbias ? a.vbias + (size_t)s * a.Cout + cw0 + 4 * lh : nullptr;
        const float* bb = a.bias ? a.bias + cw0 + 4 * lh : nullptr;
#pragma unroll
        for (int rbi = 0; rbi < 2; ++rbi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbi * 32 + (r & 3) + 8 * (r >> 2);
                if (cw0 + 4 * lh + row >= a.Cout) continue;
                float v = acc[rbi][cb][r];
                if (bb) v += bb[row];
                if (vb) v += vb[row];
                const size_t o = ob + (size_t)row * a.HW;
                if (a.res) v += a.res[o];
                ybase[o] = v;
            }
    }
}

// w [M][K] row-major (transposed = 0: M = Cout, K = Cin, w = OIHW 1x1; transposed = 1: the dgrad operand W^T, M = Cin,
// K = Cout, read from the same OIHW tensor) -> split + packed planes (layout above).  One thread per (m, k group of 8).
__global__ __launch_bounds__(256) void conv1x1_bf16x3_pack_kernel(const float* __restrict__ w, unsigned* __restrict__ out,
                                                                   int M, int K, int ldw, int transposed, int nchunks,
                                                                   int total) {
    const int idx = blockIdx.x * 256 + threadIdx.x;             // over (co tile, chunk, ks, rb, lane)
    if (idx >= total) return;
    const int lane = idx & 63, rb = (idx >> 6) & 3, ks = (idx >> 8) & 1;
    const int cc = idx >> 9;                                     // co tile * nchunks + chunk
    const int chunk = cc % nchunks, ct = cc / nchunks;
    const int m = ct * B3_TCO + rb * 32 + (lane & 31);
    const int k0 = chunk * B3_KC + ks * 16 + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = k0 + j;
        v[j] = (m < M && k < K) ? (transposed ? w[(size_t)k * ldw + m] : w[(size_t)m * ldw + k]) : 0.f;
    }
    u32x4 p1, p2, p3;
    split8(v, p1, p2, p3);
    unsigned* o = out + (((size_t)cc * 2 + ks) * 4 + rb) * (3 * 64 * 4) + lane * 4;
    *reinterpret_cast<u32x4*>(o) = p1;
    *reinterpret_cast<u32x4*>(o + 64 * 4) = p2;
    *reinterpret_cast<u32x4*>(o + 2 * 64 * 4) = p3;
}

int ilog2x(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

extern "C" {

// dwords of the packed split weights of an [M][K] operand
long vf_conv1x1_bf16x3_pack_dwords(int M, int K) {
    return (long)(rup3(M, B3_TCO) / B3_TCO) * (rup3(K, B3_KC) / B3_KC) * (2 * 4 * 3 * 64 * 4);
}

// w_oihw [Cout][Cin] (1x1) -> fwd pack (M = Cout, K = Cin) and, if w3_bwd != NULL, the dgrad pack (M = Cin, K = Cout)
int vf_conv1x1_bf16x3_pack(const float* w_oihw, void* w3_fwd, void* w3_bwd, int Cout, int Cin, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    {
        const int nch = rup3(Cin, B3_KC) / B3_KC;
        const int total = (rup3(Cout, B3_TCO) / B3_TCO) * nch * 2 * 4 * 64;
        hipLaunchKernelGGL(conv1x1_bf16x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w_oihw,
                           (unsigned*)w3_fwd, Cout, Cin, Cin, 0, nch, total);
    }
    if (w3_bwd) {
        const int nch = rup3(Cout, B3_KC) / B3_KC;
        const int total = (rup3(Cin, B3_TCO) / B3_TCO) * nch * 2 * 4 * 64;
        hipLaunchKernelGGL(conv1x1_bf16x3_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, st, w_oihw,
                           (unsigned*)w3_bwd, Cin, Cout, Cin, 1, nch, total);
    }
    VF_RETURN_LAST_ERROR();
}

// y [S][Cout][HW] = W x (+ bias[co] + view_bias[s][co] + residual); x may be the concatenation [x | x2] (x2 != NULL:
// C1in channels in x), y may be split into [y | y2] (y2 != NULL: C1out channels in y -- the dgrad of a concatenation).
// HW a power of two >= 64.  w3 from vf_conv1x1_bf16x3_pack (the fwd pack; for a dgrad pass the bwd pack with Cin / Cout
// exchanged by the caller).
int vf_conv1x1_bf16x3(const float* x, const float* x2, int C1in, const void* w3, const float* bias,
                      const float* view_bias, const float* residual, float* y, float* y2, int C1out, int S, int Cin,
                      int Cout, int HW, void* stream) {
    if (S <= 0) return 0;
    const int sh = ilog2x(HW);
    if (sh < 6 || (x2 && (C1in <= 0 || C1in >= Cin)) || (y2 && (C1out <= 0 || C1out >= Cout))) return (int)hipErrorInvalidValue;
    B3Args a;
    a.x = x; a.x2 = x2; a.w3 = (const unsigned*)w3; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y; a.y2 = y2;
    a.S = S; a.Cin = Cin; a.Cout = Cout; a.C1in = C1in; a.C1out = C1out; a.HW = HW; a.hwsh = sh; a.N = S * HW;
    a.nchunks = rup3(Cin, B3_KC) / B3_KC;
    const int nblk = ((a.N + B3_TN - 1) / B3_TN) * (rup3(Cout, B3_TCO) / B3_TCO);
    hipLaunchKernelGGL(conv1x1_bf16x3_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, a);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"
