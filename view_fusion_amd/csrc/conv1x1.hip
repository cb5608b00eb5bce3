// 1x1 convolution, forward and dgrad, at training-size grids (reference model/unet.py:238,255,256: the residual, qkv and
// output projections; round 4).  A plain GEMM  Y[co][p] = sum_ci W[co][ci] X[ci][p]  over the P = S*H*W pixels of the
// view batch, written for the rule that governs every fp32-MFMA kernel of this tree -- nothing but MFMAs is free:
//   * workgroup = 8 waves = 128 output channels x 128 pixels; wave (cw, pq) owns 64 co x 32 pixels = two
//     v_mfma_f32_32x32x2_f32 blocks (32 accumulators): every activation fragment feeds two MFMAs;
//   * the weights never touch LDS: they arrive in the packed layout of the direct kernel
//     [co tile 64][ci chunk 32][group 4][co 64][ci 8] (vf_conv_pack_weights, forward or dgrad pack) and a wave's A
//     fragments (32 rows x the lane's four k values = one contiguous 1 KB line) go STRAIGHT from global memory into the
//     MFMA A registers, reloaded in place one chunk ahead (as the Winograd kernels do with U);
//   * activations: 32 channels x 128 pixels per chunk staged global -> registers -> LDS, double buffered: ONE
//     LDS-only barrier per chunk; per chunk and wave 32 MFMAs against 8 weight loads + 16 LDS fragment reads + 2 + 2
//     staging instructions (the generic kernel: 16 MFMAs against 20 LDS reads + 6 + 6 staging instructions and two
//     barriers);
//   * every load and LDS store of the loop is unconditional (chunk / pixel indices clamped);
//   * two workgroups per CU (<= 128 VGPRs, 34 KB of LDS each): one's prologue / epilogue under the other's loop.
// The input may be the never-materialised concatenation [x | x2] (chunk-uniform source), the output the split [y | y2]
// (tile-uniform destination), as in conv.hip.  Small grids (the sampler) stay on conv_mfma_kernel + split-K.
#include "common.h"
#include "conv1x1.h"
#include <cstdlib>

namespace {

constexpr int XPS = 132;                 // LDS row stride of the activation image (128 pixels + 4: 16-byte aligned rows)
constexpr int XSZ = 32 * XPS;            // floats of one activation buffer
constexpr int VFI_C11_MINCIN64 = 128;    // fewest input channels the 64-channel variant is taken for (policy, see vfi_conv1x1_halves)
#define VF_G1 __attribute__((address_space(1)))

// NCW = 64-channel halves per workgroup: 2 (rounds 4: 128 channels x 128 pixels, 8 waves) or 1 (round 5: 64 channels x 128
// pixels, 4 waves -- the layers with an ODD number of 64-channel tiles (Cout = 64, 192, 320), where the 128-channel
// workgroup idles half the waves of its last column, and the short-K 64x64-map layers, which want more workgroups per CU
// streaming at once: four of these fit where two of the others did)
template <int NCW>
__global__ __launch_bounds__(256 * NCW, 4) void conv1x1_kernel(C11Args a) {
    __shared__ __attribute__((aligned(16))) float Xl[2 * XSZ];
    constexpr int NTH = 256 * NCW, NE = 1024 / NTH;      // staging elements (float4) per thread and chunk: 2 / 4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = NCW == 2 ? (wid & 1) : 0, pq = NCW == 2 ? (wid >> 1) : wid;
    const int li = lane & 31, lh = lane >> 5;
    const int ncp = NCW == 2 ? ((a.nct + 1) >> 1) : a.nct;   // workgroup columns of 64 NCW channels
    const unsigned logical = xcd_remap(blockIdx.x, gridDim.x);
    const int cp = logical % ncp, tp = logical / ncp;    // channel column fastest: neighbours share the pixel tile
    const int cot_raw = NCW * cp + cw;
    const bool have_co = cot_raw < a.nct;                // (odd number of 64-channel tiles: the last column's upper half idles)
    const int cot = have_co ? cot_raw : a.nct - 1;
    const int nch = a.Cin >> 5;

    // ---- activation staging: element e = tid + NTH i -> (channel e >> 5, float4 e & 31) of the 32 x 128 chunk
    unsigned xo[NE][2];                                  // byte offsets of this thread's elements in x / in x2
    int xl[NE];
    const int CA = a.x2 ? a.C1 : a.Cin, CB = a.Cin - a.C1;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + NTH * i;
        const int ci = e >> 5, q = e & 31;
        const int px = min(tp * 128 + 4 * q, a.npx - 4);             // (pixels past the batch: re-read the last ones)
        const int s = px >> a.hwsh, hw = px & (a.HW - 1);
        xo[i][0] = 4u * (unsigned)(((size_t)s * CA + ci) * a.HW + hw);
        xo[i][1] = 4u * (unsigned)(((size_t)s * CB + ci) * a.HW + hw);
        xl[i] = ci * XPS + 4 * q;
    }
    f32x4 xr0, xr1, xr2, xr3;                            // (xr2, xr3: the 64-channel variant stages four float4 per thread)
    auto fetch_x = [&](int c, f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3) {
        const int c0 = c << 5;
        const bool second = a.x2 && c0 >= a.C1;          // (uniform: the split is chunk aligned)
        const char* b_ = reinterpret_cast<const char*>(second ? a.x2 : a.x) + (size_t)(second ? c0 - a.C1 : c0) * a.HW * 4;
        unsigned o0 = second ? xo[0][1] : xo[0][0], o1 = second ? xo[1][1] : xo[1][0];
        asm("" : "+s"(b_), "+v"(o0), "+v"(o1));
        r0 = *(const VF_G1 f32x4*)((const VF_G1 char*)b_ + o0);
        r1 = *(const VF_G1 f32x4*)((const VF_G1 char*)b_ + o1);
        if constexpr (NE == 4) {
            unsigned o2 = second ? xo[NE - 2][1] : xo[NE - 2][0], o3 = second ? xo[NE - 1][1] : xo[NE - 1][0];
            asm("" : "+v"(o2), "+v"(o3));
            r2 = *(const VF_G1 f32x4*)((const VF_G1 char*)b_ + o2);
            r3 = *(const VF_G1 f32x4*)((const VF_G1 char*)b_ + o3);
        }
    };
#define VF_XSTORE(BUF, R0, R1)                                                                          \
    {                                                                                                   \
        *reinterpret_cast<f32x4*>(Xl + (BUF) * XSZ + xl[0]) = R0;                                        \
        *reinterpret_cast<f32x4*>(Xl + (BUF) * XSZ + xl[1]) = R1;                                        \
        if constexpr (NE == 4) {                                                                        \
            *reinterpret_cast<f32x4*>(Xl + (BUF) * XSZ + xl[NE - 2]) = xr2;                              \
            *reinterpret_cast<f32x4*>(Xl + (BUF) * XSZ + xl[NE - 1]) = xr3;                              \
        }                                                                                               \
    }

    // ---- weights: A fragments of (group g, 32-row block cb) of this wave's 64-channel tile, current chunk
    const char* const wbase = reinterpret_cast<const char*>(a.w + (size_t)cot * nch * (4 * 64 * 8));
    unsigned aoff = 4u * (unsigned)(li * 8 + 4 * lh);
    f32x4 a0_0, a0_1, a1_0, a1_1, a2_0, a2_1, a3_0, a3_1;
#define VF_A(G, CB) a##G##_##CB
#define VF_ALOAD(G, CB, C)                                                                              \
    {                                                                                                   \
        const char* ab_ = wbase + ((size_t)(C) * (4 * 64 * 8) + (G) * (64 * 8)) * 4;                     \
        asm("" : "+s"(ab_), "+v"(aoff));                                                                \
        VF_A(G, CB) = *(const VF_G1 f32x4*)((const VF_G1 char*)ab_ + aoff + (CB) * (32 * 8 * 4));           \
    }
#define VF_ALOAD_ALL(C)                                                                                 \
    { VF_ALOAD(0, 0, C); VF_ALOAD(0, 1, C); VF_ALOAD(1, 0, C); VF_ALOAD(1, 1, C);                        \
      VF_ALOAD(2, 0, C); VF_ALOAD(2, 1, C); VF_ALOAD(3, 0, C); VF_ALOAD(3, 1, C); }

    // B fragments: channel 8 g + 4 lh + e of pixel pq * 32 + li
    const int boff = 4 * lh * XPS + pq * 32 + li;
#define VF_BFRAG(D, BUF, G) { _Pragma("unroll") for (int e = 0; e < 4; ++e) D[e] = Xl[(BUF) * XSZ + boff + (8 * (G) + e) * XPS]; }

    f32x16 acc0 = (f32x16){0}, acc1 = (f32x16){0};
    const int clast = nch - 1;
    // one group = eight MFMAs; side work sits BEHIND MFMAs (fp32 MFMA and everything else issue serially on a SIMD)
#define VF_GROUP(G, BC, SIDE0, SIDE1)                                                                    \
    {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 0).x, BC[0], acc0, 0, 0, 0);                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        SIDE0;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 1).x, BC[0], acc1, 0, 0, 0);                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 0).y, BC[1], acc0, 0, 0, 0);                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        SIDE1;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 1).y, BC[1], acc1, 0, 0, 0);                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 0).z, BC[2], acc0, 0, 0, 0);                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 1).z, BC[2], acc1, 0, 0, 0);                    \
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 0).w, BC[3], acc0, 0, 0, 0);                    \
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(VF_A(G, 1).w, BC[3], acc1, 0, 0, 0);                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
    }
    // One chunk, PAR = its activation buffer.  Fragments of group g+1 are read behind the first MFMA of group g; the raw
    // activations of chunk C+1 (requested during chunk C-1) go to X[PAR^1] in group 0, those of chunk C+2 are requested
    // in group 1; a group's weights of chunk C+1 are requested as soon as the group is through.  The chunk's ONE barrier
    // sits at its end: X[PAR^1] is then complete, and every read of X[PAR] (rewritten in group 0 of chunk C+1) is over.
#define VF_CHUNK(C, PAR)                                                                                 \
    {                                                                                                    \
        const int cn_ = min((C) + 1, clast);                                                             \
        VF_GROUP(0, bA, VF_BFRAG(bB, PAR, 1), VF_XSTORE((PAR) ^ 1, xr0, xr1));                           \
        VF_ALOAD(0, 0, cn_); VF_ALOAD(0, 1, cn_);                                                        \
        VF_GROUP(1, bB, VF_BFRAG(bA, PAR, 2), fetch_x(min((C) + 2, clast), xr0, xr1, xr2, xr3));                   \
        VF_ALOAD(1, 0, cn_); VF_ALOAD(1, 1, cn_);                                                        \
        VF_GROUP(2, bA, VF_BFRAG(bB, PAR, 3), (void)0);                                                  \
        VF_ALOAD(2, 0, cn_); VF_ALOAD(2, 1, cn_);                                                        \
        VF_GROUP(3, bB, (void)0, (void)0);                                                               \
        VF_ALOAD(3, 0, cn_); VF_ALOAD(3, 1, cn_);                                                        \
        VF_LDS_BARRIER();                                                                                \
        VF_BFRAG(bA, (PAR) ^ 1, 0);                                                                      \
    }

    // ---- prologue: weights(0), activations(0) -> X[0], activations(1) in registers
    float bA[4], bB[4];
    VF_ALOAD_ALL(0);
    fetch_x(0, xr0, xr1, xr2, xr3);
    VF_XSTORE(0, xr0, xr1);
    fetch_x(min(1, clast), xr0, xr1, xr2, xr3);
    VF_LDS_BARRIER();
    VF_BFRAG(bA, 0, 0);
    {
        int c = 0;
        for (; c + 1 < nch; c += 2) {
            VF_CHUNK(c, 0);
            VF_CHUNK(c + 1, 1);
        }
        if (c < nch) VF_CHUNK(c, 0);
    }

    // ---- epilogue: lane = pixel (coalesced 128-byte runs per channel row), register = output channel
    const int px = tp * 128 + pq * 32 + li;
    if (!have_co || px >= a.npx) return;
    const int s = px >> a.hwsh, hw = px & (a.HW - 1);
    const bool second = a.y2 && cot * 64 >= a.C1o;       // split output: whole 64-channel tiles go to y or to y2
    float* const yout = second ? a.y2 : a.y;
    const int CY = second ? a.Cout - a.C1o : (a.y2 ? a.C1o : a.Cout);
    const int coY = cot * 64 - (second ? a.C1o : 0) + 4 * lh;
    const int cob = cot * 64 + 4 * lh;                   // channel of register 0 in the conv's own numbering
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const f32x16& acc = cb == 0 ? acc0 : acc1;
        // every operand of the block is requested before its first store (clamped channel indices, one uniform branch
        // per operand kind)
        float add[16], ad2[16], ad3[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) add[r] = ad2[r] = ad3[r] = 0.f;
        const size_t ob = ((size_t)s * CY + coY + cb * 32) * a.HW + hw;
        if (a.res) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dco = min((r & 3) + 8 * (r >> 2), a.Cout - 1 - cob - cb * 32);
                add[r] = a.res[ob + (size_t)max(dco, 0) * a.HW];
            }
        }
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ad2[r] = a.bias[min(cob + cb * 32 + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
        }
        if (a.vbias) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                ad3[r] = a.vbias[(size_t)s * a.Cout + min(cob + cb * 32 + (r & 3) + 8 * (r >> 2), a.Cout - 1)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dco = (r & 3) + 8 * (r >> 2);
            if (cob + cb * 32 + dco < a.Cout) yout[ob + (size_t)dco * a.HW] = acc[r] + (add[r] + (ad2[r] + ad3[r]));
        }
    }
#undef VF_XSTORE
#undef VF_ALOAD
#undef VF_A
#undef VF_ALOAD_ALL
#undef VF_BFRAG
#undef VF_GROUP
#undef VF_CHUNK
}

}  // namespace

// Where it is taken (measured against conv_mfma_kernel over the 17 shapes of tools/conv1x1_table.py, S = 96, round 4):
// it wins where the K loop is deep and the 128-channel workgroup columns are full -- 256 -> 128 @32x32 58.5 vs 69.1 us,
// 320 -> 128 @32x32 68.4 vs 82.7 (110-118 TF = 0.70-0.75 of the fp32 MFMA peak) -- and loses where they are not: an odd
// number of 64-channel tiles idles half of the last column's waves (Cout = 64: 144 vs 78 us; 192, 320, 576), a short K
// (Cin = 64, 128: two to four chunks) leaves the workgroup's 64 KB output store and first loads uncovered (the
// 64x64-map layers are at 3.9 TB/s of HBM traffic with the generic kernel already), and 128 x 128 tiles give the 16x16
// maps too few workgroups (24576 pixels per layer at S = 96).  So: whole 32-channel chunks from either source,
// tile-aligned output split, at least eight chunks, an even number of 64-channel tiles, two workgroups per CU.
int vfi_conv1x1_halves(const C11Args& a);

bool vfi_conv1x1_supported(const C11Args& a) {
    if (a.Cin % 32 != 0 || a.HW < 64 || (a.HW & (a.HW - 1)) != 0 || a.npx < 128 || a.npx % 4 != 0) return false;
    if (a.x2 && (a.C1 <= 0 || a.C1 >= a.Cin || a.C1 % 32 != 0)) return false;
    if (a.y2 && (a.C1o <= 0 || a.C1o >= a.Cout || a.C1o % 64 != 0)) return false;
    // the kernel carries per-lane BYTE offsets into x / x2 as 32-bit values: a source tensor must stay below 4 GiB
    // (S = 96 views of 320 channels at 64x64 are 0.5 GiB; larger view batches fall back to the generic kernel)
    const size_t widest = (size_t)(a.x2 ? (a.C1 > a.Cin - a.C1 ? a.C1 : a.Cin - a.C1) : a.Cin);
    if ((size_t)a.S * widest * (size_t)a.HW * 4 >= (1ull << 32)) return false;
    static const bool force = getenv("VF_CONV1X1_FORCE") != nullptr;     // tests / tuning: wherever the shape is legal
    const long tiles = (long)((a.npx + 127) / 128) * ((a.nct + 1) / 2);
    if (force) return tiles >= 64;
    return vfi_conv1x1_halves(a) != 0;
}

// 2: the 128-channel workgroup, 1: the 64-channel one, 0: the generic kernel.  VF_CONV1X1_NCW = 1 | 2 forces a variant
// wherever the kernel is taken at all (tuning aid).
int vfi_conv1x1_halves(const C11Args& a) {
    static const int forced = getenv("VF_CONV1X1_NCW") ? atoi(getenv("VF_CONV1X1_NCW")) : 0;
    if (forced == 1 || forced == 2) return forced;
    static const bool force = getenv("VF_CONV1X1_FORCE") != nullptr;
    const long ptiles = (a.npx + 127) / 128;
    if (a.Cin >= 256 && (a.nct & 1) == 0 && ptiles * (a.nct / 2) >= 512) return 2;      // round 4's rule
    if (force) return (a.nct & 1) ? 1 : 2;
    static const bool on64 = !(getenv("VF_CONV1X1_64") && getenv("VF_CONV1X1_64")[0] == '0');
    // round 5 (tools/conv1x1_table.py, profiles/r05_conv1x1_variants.txt): the 64-channel workgroup where the 128-channel one
    // does not apply AND it beats the generic kernel: at least four 64-channel tiles and K >= 128 -- 192 -> 576 @16x16 69.5
    // -> 63.0 us, 192 -> 384 49.8 -> 44.6, 128 -> 256 @32x32 71.3 -> 68.4, the 8x8 layers with 320 ... 960 channels -4 %; it
    // loses with three tiles (Cout = 192 @16x16: 60.5 -> 65.0) and on the short-K 64x64 layers (64 -> 128: 97 -> 108)
    if (on64 && a.Cin >= VFI_C11_MINCIN64 && a.nct >= 4 && ptiles * a.nct >= 192) return 1;
    return 0;
}

int vfi_conv1x1_launch(const C11Args& a, hipStream_t st) {
    const int halves = vfi_conv1x1_halves(a) == 1 ? 1 : 2;
    const unsigned ptiles = (unsigned)((a.npx + 127) / 128);
    if (halves == 2) hipLaunchKernelGGL(conv1x1_kernel<2>, dim3(ptiles * (unsigned)((a.nct + 1) / 2)), dim3(512), 0, st, a);
    else hipLaunchKernelGGL(conv1x1_kernel<1>, dim3(ptiles * (unsigned)a.nct), dim3(256), 0, st, a);
    VF_RETURN_LAST_ERROR();
}
