// Winograd F(2x2, 3x3) WEIGHT GRADIENT of the stride-1 3x3 layers (reference model/unet.py:42,189,214; the forward
// and dgrad passes of these layers run the nested F(2,3) x F(4,3) kernel of winograd24.hip -- the weight gradient
// takes x and dY directly, so the two transforms are independent).
#include "common.h"

namespace {
inline int rup(int v, int m) { return (v + m - 1) / m * m; }
}  // namespace

// =================================================================================================
// Winograd weight gradient:  dU_k[co][ci] = sum_tiles (A dY A^T)_k[co][tile] * (B^T d B)_k[ci][tile],
// dW = G^T dU G (applied by the reduce kernel after the split-K slabs are summed, fixed order).
// Same 8-wave structure as the forward kernel: wave (cw, ciw, kh) owns 32 co x 32 ci x 8 slices = 128
// accumulators; K = tiles, 8 tiles (one 16-pixel x 2-row strip) per chunk; both MFMA operands come
// from LDS by ds_read_b128 (4 consecutive tiles per lane half).  dY tiles are transformed straight
// from global registers, x goes through a raw LDS strip (neighbouring windows overlap by 2 pixels).
namespace {

constexpr int GT = 8;                 // tiles per chunk

struct WinoWgradArgs {
    const float* x;
    const float* dy;
    float* ws;                        // [slab][k 16][CoutP][CinQ]
    float* bsum;                      // [slab][CoutP] per-slice sums of dY per output channel (bias gradient), or null
    int S, Cin, Cout, CoutP, CinQ;
    int nchunks, chunks_per_slice;
};

template <int LOGW, int MODE>
__global__ __launch_bounds__(512, 2) void wino_wgrad_kernel(WinoWgradArgs a) {
    constexpr int W = 1 << LOGW, H = W;
    constexpr int SW = MODE == 2 ? W / 2 : W, SH = SW;      // stored input size (MODE 2: nearest-upsampled x2 on read)
    constexpr int TPR = W / 2 < GT ? W / 2 : GT;      // tiles of one tile row inside a chunk
    constexpr int TR = GT / TPR;                      // tile rows per chunk (2 on 8x8 maps, else 1)
    constexpr int NR = 2 * TR + 2;                    // input rows of the raw strip
    constexpr int QPR = TPR / 2;                      // float4 per strip row
    constexpr int GXW = 2 * TPR + 8;                  // strip row: idx 3 = left halo, 4.. = pixels, 4+2*TPR = right halo
    constexpr bool HALO = W > 2 * TPR;                // strip narrower than the map: halo columns carry data
    constexpr int NX4 = 64 * NR * QPR;                // float4 of a strip (<= 1024)
    constexpr int CPR = (W / 2) / TPR;                // chunks per (group of TR) tile rows
    constexpr int CPI = (H / 2) / TR * CPR;           // chunks per image
    constexpr int MSZ = 16 * 64 * GT;                 // floats of one dM (or V) buffer
    constexpr int XCS = NR * GXW + 16;                // strip stride per channel = 16 mod 32 banks: window reads 2-way, not 4-way
    constexpr int XSZ = 64 * XCS;

    __shared__ __attribute__((aligned(16))) float lds[4 * MSZ + XSZ];
    float* const Ml = lds;                            // dM[2][k][co][tile]
    float* const Vl = lds + 2 * MSZ;                  // V [2][k][ci][tile]
    float* const Xl = lds + 4 * MSZ;                  // raw x strip [ci][NR rows][GXW]

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, ciw = (wid >> 1) & 1, kh = wid >> 2;
    const int li = lane & 31, lh = lane >> 5;
    // XCD-aware decode of (co tile, ci tile, K slice): the workgroups of ONE slice stream the same x and dY chunks
    // (x is shared by the co tiles, dY by the ci tiles).  Hardware deals linear block ids round-robin over the 8 XCDs,
    // each with its own L2; xcd_remap makes consecutive LOGICAL ids -- the pairs of one slice -- run on the same XCD
    // at the same time, so a chunk is fetched from HBM once per slice instead of once per XCD that touches it.
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned lgc = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
    const int bx = lgc % gridDim.x, by = (lgc / gridDim.x) % gridDim.y, bz = lgc / (gridDim.x * gridDim.y);
    const int co0 = bx * 64, ci0 = by * 64;
    const int c_begin = bz * a.chunks_per_slice;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_slice);
    const int n = c_end - c_begin;

    // this thread's transform duty: channel tch (a co for dM, a ci for V) and tile tt of the chunk
    const int tch = tid >> 3, tt = tid & 7;
    const int ttr = tt / TPR, ttc = tt % TPR;         // tile row / column inside the chunk
    const int tsw = tt ^ (((tch >> 4) & 1) << 2);     // LDS slot: 16-B halves swapped on rows 16-31, 48-63 (b128 banks)

    // fp32 MFMAs and VALU instructions do not overlap on a SIMD (tools/mfma_valu.hip), so the chunk loop carries as
    // little vector arithmetic as possible:
    //  * every global address is a workgroup-uniform base (SGPR pair, moved per chunk by scalar code) + a 32-bit byte
    //    offset that is a per-thread CONSTANT -- the saddr form of global_load, no address VALU;
    //  * rows / columns outside the image are not loaded (uniform-and-constant predicates = scalar exec masks); their
    //    LDS slots get zeros from a branch only the edge chunks take;
    //  * both transforms run on packed fp32 (v_pk_add_f32 with op_sel / neg modifiers);
    //  * the chunk loop is unrolled by buffer parity: every LDS address is a per-thread base + an immediate.
    // raw x staging duty: NX4 float4 (<= 2 per thread), e -> (ci, row, q) = (e / (NR*QPR), (e / QPR) % NR, e % QPR);
    // with HALO 64*4*2 = 512 halo scalars, one per thread: (ci, row, side) = (tid >> 3, (tid >> 1) & 3, tid & 1)
    unsigned xoffb[2];
    int xsl[2];
    bool xok[2], xtop[2], xbot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + 512 * i;
        const int ci = e / (NR * QPR), row = (e / QPR) % NR, q = e % QPR;
        xok[i] = e < NX4 && ci0 + ci < a.Cin;
        xtop[i] = row == 0;
        xbot[i] = row == NR - 1;
        // source row gy = 2p - 1 + row (MODE 2: stored row (gy >> 1) = p - 1 + ((row + 1) >> 1)); the chunk-dependent
        // part, including the "- 1", lives in the uniform base
        xoffb[i] = 4u * (unsigned)(MODE == 2 ? ci * (SH * SW) + ((row + 1) >> 1) * SW + 2 * q : ci * (H * W) + row * W + 4 * q);
        xsl[i] = ci * XCS + row * GXW + 4 + 4 * q;
    }
    const int hci = tid >> 3, hrow = (tid >> 1) & 3, hside = tid & 1;
    const bool hok = HALO && ci0 + hci < a.Cin, htop = hrow == 0, hbot = hrow == NR - 1;
    // halo pixel gx = 2 q0 - 1 (left) or 2 q0 + 2 TPR (right); the "2 q0 - 1" is in the base
    unsigned hoffb = 4u * (unsigned)(MODE == 2 ? hci * (SH * SW) + ((hrow + 1) >> 1) * SW + (hside ? TPR + 1 : 0)
                                                     : hci * (H * W) + hrow * W + (hside ? 2 * TPR + 1 : 0));
    const int hsl = hci * XCS + hrow * GXW + (hside ? 4 + 2 * TPR : 3);
    const bool dyok = co0 + tch < a.Cout;
    unsigned dyoffb = 4u * (unsigned)(tch * (H * W) + 2 * ttr * W + 2 * ttc);

    // the predicates as wave masks in SGPRs: "uniform flag AND per-thread constant" is scalar mask arithmetic, applied
    // to the exec mask directly (inverse ballot) -- written with bools the compiler evaluates it per lane in VALU
    typedef unsigned long long mask_t;
    const mask_t m_x[2] = {__builtin_amdgcn_ballot_w64(xok[0]), __builtin_amdgcn_ballot_w64(xok[1])};
    const mask_t m_xtop[2] = {__builtin_amdgcn_ballot_w64(xok[0] && xtop[0]), __builtin_amdgcn_ballot_w64(xok[1] && xtop[1])};
    const mask_t m_xbot[2] = {__builtin_amdgcn_ballot_w64(xok[0] && xbot[0]), __builtin_amdgcn_ballot_w64(xok[1] && xbot[1])};
    const mask_t m_h = __builtin_amdgcn_ballot_w64(hok), m_htop = __builtin_amdgcn_ballot_w64(hok && htop),
                 m_hbot = __builtin_amdgcn_ballot_w64(hok && hbot), m_hleft = __builtin_amdgcn_ballot_w64(hok && !hside),
                 m_hright = __builtin_amdgcn_ballot_w64(hok && hside);
    const mask_t m_dy = __builtin_amdgcn_ballot_w64(dyok);
#define VF_LANES(M) __builtin_amdgcn_inverse_ballot_w64(M)

    f32x4 xr0 = (f32x4){0.f, 0.f, 0.f, 0.f}, xr1 = xr0;
    float xh = 0.f;
    f32x2 dy0 = (f32x2){0.f, 0.f}, dy1 = dy0;

    struct Chunk {                                     // uniform: position of a chunk and its edge flags
        const char* xb;                                // base of the x loads (points one row / pixel before the strip)
        const char* hb;
        const char* db;
        bool top, bot, left, right;
    };
    auto chunk_at = [&](int c) -> Chunk {
        const int s = c / CPI;
        const int r = c - s * CPI;
        const int p = (r / CPR) * TR;                  // first tile row
        const int q0 = (r % CPR) * TPR;                // first tile column
        Chunk k;
        const long long img = ((long long)s * a.Cin + ci0) * (SH * SW);
        const long long xo = MODE == 2 ? img + (long long)(p - 1) * SW + q0 : img + (long long)(2 * p - 1) * W + 2 * q0;
        k.xb = uniform_ptr(reinterpret_cast<const char*>(a.x) + 4 * xo);
        k.hb = k.xb - 4;
        k.db = uniform_ptr(reinterpret_cast<const char*>(a.dy) + 4 * ((((long long)s * a.Cout + co0) * H + 2 * p) * W + 2 * q0));
        k.top = p == 0;
        k.bot = p + TR == H / 2;
        k.left = q0 == 0;
        k.right = q0 + TPR == W / 2;
        return k;
    };
#define VF_G1 __attribute__((address_space(1)))
    // (the opaque copies keep base and offset apart until instruction selection: SGPR base + zero-extended VGPR offset)
    auto load_x = [&](const Chunk& k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const mask_t edge = (k.top ? m_xtop[i] : 0ull) | (k.bot ? m_xbot[i] : 0ull);
            if (VF_LANES(m_x[i] & ~edge)) {
                const char* b = k.xb;
                unsigned& o = xoffb[i];
                asm("" : "+s"(b), "+v"(o));
                if (MODE == 2) {
                    const f32x2 h = *(const VF_G1 f32x2*)((const VF_G1 char*)b + o);
                    if (i == 0) xr0 = (f32x4){h.x, h.x, h.y, h.y}; else xr1 = (f32x4){h.x, h.x, h.y, h.y};
                } else {
                    if (i == 0) xr0 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o);
                    else xr1 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o);
                }
            }
        }
        if (HALO) {
            const mask_t edge = (k.top ? m_htop : 0ull) | (k.bot ? m_hbot : 0ull) | (k.left ? m_hleft : 0ull) |
                                (k.right ? m_hright : 0ull);
            if (VF_LANES(m_h & ~edge)) {
                const char* hb = k.hb;
                unsigned& o = hoffb;
                asm("" : "+s"(hb), "+v"(o));
                xh = *(const VF_G1 float*)((const VF_G1 char*)hb + o);
            }
        }
    };
    auto store_x = [&](const Chunk& k) {               // k: the chunk the registers hold
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const mask_t edge = (k.top ? m_xtop[i] : 0ull) | (k.bot ? m_xbot[i] : 0ull);
            f32x4* dst = reinterpret_cast<f32x4*>(Xl + xsl[i]);
            // (the empty asm keeps the rarely taken zero store a branch: merged with the data store it becomes four
            // selects per chunk)
            if (VF_LANES(edge)) { asm volatile(""); *dst = (f32x4){0.f, 0.f, 0.f, 0.f}; }
            if (VF_LANES(m_x[i] & ~edge)) *dst = i == 0 ? xr0 : xr1;
        }
        if (HALO) {
            const mask_t edge = (k.top ? m_htop : 0ull) | (k.bot ? m_hbot : 0ull) | (k.left ? m_hleft : 0ull) |
                                (k.right ? m_hright : 0ull);
            if (VF_LANES(edge)) { asm volatile(""); Xl[hsl] = 0.f; }
            if (VF_LANES(m_h & ~edge)) Xl[hsl] = xh;
        }
    };
    // the strip is zeroed once: channels beyond Cin stay zero, and (no HALO) so do both halo columns
    for (int e = tid; e < XSZ; e += 512) Xl[e] = 0.f;
    auto load_dy = [&](const Chunk& k) {
        if (VF_LANES(m_dy)) {
            const char* b = k.db;
            unsigned& o = dyoffb;
            asm("" : "+s"(b), "+v"(o));
            dy0 = *(const VF_G1 f32x2*)((const VF_G1 char*)b + o);
            dy1 = *(const VF_G1 f32x2*)((const VF_G1 char*)b + o + 4 * W);
        }
    };
    // dM' = A' dY A'^T with A'^T = [[1,1,1,0],[0,1,-1,1]]: the transpose of the output transform with its last row
    // POSITIVE (no negations here); wino_wgrad_finish_kernel flips the sign of the slices (3, j < 3) and (i < 3, 3).
    f32x2 bias2 = (f32x2){0.f, 0.f};                   // sums of this thread's dY tiles (the bias gradient rides along)
    auto xform_dy = [&](int buf, bool count) {         // count (uniform): false for the clamped repeats of the last chunk
        const f32x2 r1 = pk_add(dy0, dy1), r2 = pk_sub(dy0, dy1);
        if (count) bias2 = pk_add(bias2, r1);
        const f32x2 u0 = pk_lo_pm_hi(dy0, dy0), u1 = pk_lo_pm_hi(r1, r1), u2 = pk_lo_pm_hi(r2, r2), u3 = pk_lo_pm_hi(dy1, dy1);
        float* mo = Ml + buf * MSZ + tch * GT + tsw;
        mo[0 * 64 * GT] = dy0.x; mo[1 * 64 * GT] = u0.x; mo[2 * 64 * GT] = u0.y; mo[3 * 64 * GT] = dy0.y;
        mo[4 * 64 * GT] = r1.x;  mo[5 * 64 * GT] = u1.x; mo[6 * 64 * GT] = u1.y; mo[7 * 64 * GT] = r1.y;
        mo[8 * 64 * GT] = r2.x;  mo[9 * 64 * GT] = u2.x; mo[10 * 64 * GT] = u2.y; mo[11 * 64 * GT] = r2.y;
        mo[12 * 64 * GT] = dy1.x; mo[13 * 64 * GT] = u3.x; mo[14 * 64 * GT] = u3.y; mo[15 * 64 * GT] = dy1.y;
    };
    // V = B^T d B, B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]: rows as packed pairs (columns 0,1 / 2,3), then per
    // row  [v0, v3] = [c0 - c2, c1 - c3]  and  [v1, v2] = [c2 + c1, c2 - c1]:  16 packed instructions per window
    f32x2 da[4], db_[4];                               // window rows: columns (0,1) and (2,3)
    auto xform_x_read = [&](int r) {
        const float* p = Xl + tch * XCS + (2 * ttr + r) * GXW + 2 * ttc + 3;
        da[r] = (f32x2){p[0], p[1]};
        db_[r] = (f32x2){p[2], p[3]};
    };
    auto xform_x_write = [&](int buf) {
        float* vo = Vl + buf * MSZ + tch * GT + tsw;
        const f32x2 ta[4] = {pk_sub(da[0], da[2]), pk_add(da[1], da[2]), pk_sub(da[2], da[1]), pk_sub(da[1], da[3])};
        const f32x2 tb[4] = {pk_sub(db_[0], db_[2]), pk_add(db_[1], db_[2]), pk_sub(db_[2], db_[1]), pk_sub(db_[1], db_[3])};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 v03 = pk_sub(ta[r], tb[r]), v12 = pk_lo_pm_hi(tb[r], ta[r]);
            vo[(4 * r + 0) * 64 * GT] = v03.x;
            vo[(4 * r + 1) * 64 * GT] = v12.x;
            vo[(4 * r + 2) * 64 * GT] = v12.y;
            vo[(4 * r + 3) * 64 * GT] = v03.y;
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (f32x16){0};

    const int clast = c_end - 1;
    if (n > 0) {
        // ---- prologue: chunk 0 transformed into buffer 0; chunk 1 raw data in registers
        const Chunk k0 = chunk_at(c_begin), k1 = chunk_at(min(c_begin + 1, clast));
        load_x(k0);
        load_dy(k0);
        __syncthreads();                               // strip zero fill done
        store_x(k0);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) xform_x_read(r);
        xform_x_write(0);
        xform_dy(0, true);
        load_x(k1);
        load_dy(k1);
        __syncthreads();
    }

    const int aoff = 8 * kh * 64 * GT + (cw * 32 + li) * GT + 4 * (lh ^ ((li >> 4) & 1));
    const int boff = 8 * kh * 64 * GT + (ciw * 32 + li) * GT + 4 * (lh ^ ((li >> 4) & 1));
    // One chunk, PAR = its buffer parity (compile time).  Side work of chunk C+1 (raw strip -> LDS in slice 0, barrier,
    // transforms in slices 4-7) and the loads of chunk C+2, interleaved with the slices' own MFMAs (the two waves of a
    // SIMD run in phase: side work in front of the MFMAs would idle the matrix pipe in both at once).  Chunk indices
    // beyond the slice are clamped: the last iterations redo harmless loads / LDS writes that nobody reads.
#define VF_WG_CHUNK(C, PAR)                                                                               \
    {                                                                                                     \
        const Chunk kn1 = chunk_at(min((C) + 1, clast)), kn2 = chunk_at(min((C) + 2, clast));             \
        const float* ab = Ml + (PAR) * MSZ + aoff;                                                        \
        const float* bb = Vl + (PAR) * MSZ + boff;                                                        \
        _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                   \
            f32x4 a_nxt, b_nxt;                                                                           \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc[k], 0, 0, 0);             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if (k + 1 < 8) {                                                                              \
                a_nxt = *reinterpret_cast<const f32x4*>(ab + (k + 1) * 64 * GT);                          \
                b_nxt = *reinterpret_cast<const f32x4*>(bb + (k + 1) * 64 * GT);                          \
            } else {                     /* behind the barrier: slice 0 of chunk C+1, from the other buffers */ \
                a_nxt = *reinterpret_cast<const f32x4*>(Ml + ((PAR) ^ 1) * MSZ + aoff);                   \
                b_nxt = *reinterpret_cast<const f32x4*>(Vl + ((PAR) ^ 1) * MSZ + boff);                   \
            }                                                                                             \
            if (k == 0) store_x(kn1);                                                                     \
            if (k == 1) load_x(kn2);                                                                      \
            if (k == 4) { __syncthreads(); xform_x_read(0); }   /* raw strip of chunk C+1 visible */        \
            if (k == 5) xform_x_read(2);                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc[k], 0, 0, 0);             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if (k == 4) xform_x_read(1);                                                                  \
            if (k == 5) xform_x_read(3);                                                                  \
            if (k == 6) xform_x_write((PAR) ^ 1);                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc[k], 0, 0, 0);             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            if (k == 5) xform_dy((PAR) ^ 1, (C) + 1 <= clast);                                            \
            if (k == 6) load_dy(kn2);                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc[k], 0, 0, 0);             \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            /* the chunk's second barrier sits behind slice 6: the images of chunk C+1 are complete, every read  \
               of this chunk's images is issued (slice 7's fragments are in registers), and slice 7 fetches the  \
               first fragments of chunk C+1 -- no wave waits for LDS at a chunk boundary */                      \
            if (k == 6) __syncthreads();                                                                  \
            a_cur = a_nxt; b_cur = b_nxt;                                                                 \
        }                                                                                                 \
    }
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(Ml + aoff);
    f32x4 b_cur = *reinterpret_cast<const f32x4*>(Vl + boff);
    {
        int c = c_begin;
        for (; c + 1 < c_end; c += 2) {
            VF_WG_CHUNK(c, 0);
            VF_WG_CHUNK(c + 1, 1);
        }
        if (c < c_end) VF_WG_CHUNK(c, 0);
    }
#undef VF_WG_CHUNK
#undef VF_G1
#undef VF_LANES
    const float bias_acc = bias2.x + bias2.y;

    // partial dU of this slice: rows k = 8kh .. 8kh+7
    const int slab = bz;
    if (a.bsum && by == 0) {                   // bias gradient partial: the 8 tile lanes of a channel, fixed order
        float b = bias_acc;
        b += __shfl_xor(b, 1, 64);
        b += __shfl_xor(b, 2, 64);
        b += __shfl_xor(b, 4, 64);
        if (tt == 0) a.bsum[(size_t)slab * a.CoutP + co0 + tch] = b;
    }
    // (one 64-bit base per workgroup lane, 32-bit offsets inside the slab: 128 stores without 64-bit multiplies)
    const int ci = ci0 + ciw * 32 + li;
    if (ci < a.CinQ) {
        float* sb = a.ws + (size_t)slab * 16 * a.CoutP * a.CinQ + (size_t)(co0 + cw * 32 + 4 * lh) * a.CinQ + ci;
        const int kst = a.CoutP * a.CinQ;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) sb[(8 * kh + k) * kst + ((r & 3) + 8 * (r >> 2)) * a.CinQ] = acc[k][r];
    }
}

// slab sum: one workgroup per (64 consecutive (co,ci) entries, slice k); 4 slab groups per entry, combined
// through LDS in a fixed order.  out = dU[k][CoutP*CinQ]
__global__ __launch_bounds__(256) void wino_wgrad_slabsum_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                                 int nslab, long kstride) {
    __shared__ float part[4][64];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6, k = blockIdx.y;
    const long off = (long)k * kstride + (long)blockIdx.x * 64 + e;
    const long sstride = 16 * kstride;
    // 8 independent loads in flight per thread (fixed summation order: slabs g, g+4, ... in 8 interleaved chains)
    float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int s = g;
    for (; s + 28 < nslab; s += 32) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = ws[(s + 4 * j) * sstride + off];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] += t[j];
    }
    {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = s + 4 * j < nslab ? ws[(s + 4 * j) * sstride + off] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc8[j] += t[j];
    }
    const float a0 = (acc8[0] + acc8[1]) + (acc8[2] + acc8[3]), a1 = (acc8[4] + acc8[5]) + (acc8[6] + acc8[7]);
    part[g][e] = a0 + a1;
    __syncthreads();
    if (g == 0) out[off] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

// dW[co][ci][p][q] = sum_{i,j} G[i][p] G[j][q] dU[4i+j][co][ci]
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ du, float* __restrict__ dw,
                                                                int Cout, int Cin, int CoutP, int CinQ,
                                                                const float* __restrict__ bsum, float* __restrict__ db,
                                                                int nslab, int nmain, float* db2) {
    if ((int)blockIdx.x >= nmain) {      // trailing blocks: db[co] = sum over the slices' dY sums (64 channels x 4
        __shared__ float red[4][64];     // slice groups per block, 8 loads in flight, fixed order)
        const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int co = ((int)blockIdx.x - nmain) * 64 + cx;
        float acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (co < Cout) {
            for (int z = g; z < nslab; z += 32) {
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = bsum[(size_t)min(z + 4 * j, nslab - 1) * CoutP + co];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc8[j] += z + 4 * j < nslab ? t[j] : 0.f;
            }
        }
        red[g][cx] = ((acc8[0] + acc8[1]) + (acc8[2] + acc8[3])) + ((acc8[4] + acc8[5]) + (acc8[6] + acc8[7]));
        __syncthreads();
        if (g == 0 && co < Cout) {
            const float v = (red[0][cx] + red[1][cx]) + (red[2][cx] + red[3][cx]);
            db[co] = v;
            if (db2) db2[co] = v;                 // a second owner (the residual 1x1 conv) gets its own copy
        }
        return;
    }
    const int idx = blockIdx.x * 256 + threadIdx.x;           // over (co, ci), ci fastest
    if (idx >= Cout * Cin) return;
    const int ci = idx % Cin, co = idx / Cin;
    const size_t kstride = (size_t)CoutP * CinQ;
    const float* p = du + (size_t)co * CinQ + ci;
    float u[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {                        // (the wgrad kernel's dY transform keeps its last row positive:
        const float v = p[k * kstride];                   //  slices (3, j < 3) and (i < 3, 3) come with the opposite sign)
        u[k] = ((k >> 2) == 3) != ((k & 3) == 3) ? -v : v;
    }
    // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] : rows first, then columns
    float t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float h1 = 0.5f * (u[4 + j] + u[8 + j]), h2 = 0.5f * (u[4 + j] - u[8 + j]);
        t[0][j] = u[j] + h1;
        t[1][j] = h2;
        t[2][j] = h1 + u[12 + j];
    }
    float* o = dw + ((size_t)co * Cin + ci) * 9;
#pragma unroll
    for (int pp = 0; pp < 3; ++pp) {
        const float h1 = 0.5f * (t[pp][1] + t[pp][2]), h2 = 0.5f * (t[pp][1] - t[pp][2]);
        o[pp * 3 + 0] = t[pp][0] + h1;
        o[pp * 3 + 1] = h2;
        o[pp * 3 + 2] = h1 + t[pp][3];
    }
}

template <int LOGW, int MODE>
int launch_wino_wgrad(WinoWgradArgs a, float* dw, float* db, float* db2, size_t ws_floats, hipStream_t st) {
    constexpr int W = 1 << LOGW;
    a.nchunks = a.S * ((W / 2) * (W / 2) / GT);
    const int nco = a.CoutP / 64, nci = (a.CinQ + 63) / 64;
    const size_t slab_floats = (size_t)16 * a.CoutP * a.CinQ;
    int z = 256 / (nco * nci);
    if (z < 1) z = 1;
    if (z > a.nchunks) z = a.nchunks;
    if (ws_floats < 2 * slab_floats + 256 * (size_t)a.CoutP) return (int)hipErrorInvalidValue;
    const size_t zmax = (ws_floats - 256 * (size_t)a.CoutP) / slab_floats - 1;
    if ((size_t)z > zmax) z = (int)zmax;
    a.chunks_per_slice = (a.nchunks + z - 1) / z;
    z = (a.nchunks + a.chunks_per_slice - 1) / a.chunks_per_slice;
    // the summed dU goes behind the slabs (vf_wino_wgrad_ws_floats reserves one extra slab), the per-slice dY sums
    // behind that
    float* du = a.ws + (size_t)z * slab_floats;
    a.bsum = db ? du + slab_floats : nullptr;
    hipLaunchKernelGGL((wino_wgrad_kernel<LOGW, MODE>), dim3(nco, nci, z), dim3(512), 0, st, a);
    const long kstride = (long)a.CoutP * a.CinQ;
    hipLaunchKernelGGL(wino_wgrad_slabsum_kernel, dim3((unsigned)(kstride / 64), 16), dim3(256), 0, st, a.ws, du, z,
                       kstride);
    const int total = a.Cout * a.Cin;
    const int nmain = (total + 255) / 256, nbias = db ? (a.Cout + 63) / 64 : 0;
    hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3(nmain + nbias), dim3(256), 0, st, du, dw, a.Cout, a.Cin, a.CoutP,
                       a.CinQ, a.bsum, db, z, nmain, db2);
    VF_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

// workspace floats for vf_wino_wgrad at this shape (slabs of transformed partial gradients)
long vf_wino22_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const long slab = 16L * rup(Cout, 64) * rup(Cin, 32);
    const int nco = rup(Cout, 64) / 64, nci = (rup(Cin, 32) + 63) / 64;
    long z = 256 / (nco * nci);
    if (z < 1) z = 1;
    const long nchunks = (long)S * (H / 2) * (W / 2) / GT;
    if (z > nchunks) z = nchunks;
    return (z + 1) * slab + 256L * rup(Cout, 64);
}


// dw[Cout][Cin][3][3] of a stride-1 3x3 conv (H = W = output size in {8, 16, 32, 64}; mode 2: x is stored at half
// size and nearest-upsampled on read) via Winograd F(2x2,3x3)
// db (or NULL): also the bias gradient sum_{s,p} dY[s][co][p] -- the kernel reads every dY tile anyway;
// db2 (or NULL): a second [Cout] destination for the same sums (the residual 1x1 conv shares this dY)
int vf_wino22_wgrad(const float* x, const float* dy, float* dw, float* db, float* db2, float* ws, long ws_floats, int S,
                  int Cin, int Cout, int H, int W, int mode, void* stream) {
    if (S <= 0) return 0;
    WinoWgradArgs a;
    a.x = x; a.dy = dy; a.ws = ws; a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CoutP = rup(Cout, 64); a.CinQ = rup(Cin, 32);
    hipStream_t st = (hipStream_t)stream;
#define VF_WG(LW) \
    return mode == 2 ? launch_wino_wgrad<LW, 2>(a, dw, db, db2, (size_t)ws_floats, st) \
                     : launch_wino_wgrad<LW, 0>(a, dw, db, db2, (size_t)ws_floats, st)
    if (W == 8) VF_WG(3);
    if (W == 16) VF_WG(4);
    if (W == 32) VF_WG(5);
    VF_WG(6);
#undef VF_WG
}

}  // extern "C"
