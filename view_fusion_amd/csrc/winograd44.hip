// Winograd F(4x4, 3x3) WEIGHT GRADIENT of the stride-1 3x3 layers (reference model/unet.py:42,189,214):
//
//     dW (3x3) = G4^T [ sum_tiles (A4 dY A4^T) (.) (B4^T d B4) ] G4       dY = 4x4 output-gradient tile, d = 6x6 window
//
// 36 products per 16 pixels and (co, ci) pair = 2.25 multiplies per pixel instead of 9 (direct) or 4 (the F(2x2,3x3)
// kernel of rounds 1-2, experiments/winograd_f22_wgrad.hip.txt).  fp32 throughout.  Error of the resulting dW against an fp64 correlation
// (random data, K = 1536 ... 24576 tiles): rel-L2 7e-6 (direct fp32 accumulation: 3e-6, F(2x2): 2e-6) -- the sum over
// tiles runs in the transformed domain, the 4-point constants (4, 8, 1/6, 1/24) enter once per axis; the tolerance
// of the gradient tests is 1e-4.  (The forward / dgrad kernel keeps the milder nested F(2,3)xF(4,3): an activation
// error feeds every later layer, a weight-gradient error does not.)
//
// One workgroup = 8 waves = 64 co x 32 ci x 36 slices; wave (cw, sg) owns 32 co x 32 ci x a 3x3 block of the 6x6
// slice grid = 144 accumulators.  K = tiles, 4 tiles (one 16-pixel x 4-row strip; 2x2 tiles on 8x8 maps) per chunk = 2 MFMAs
// (v_mfma_f32_32x32x2_f32) per slice; both operands are ds_read_b64 fragments of the images
//     dM[slice][co 64][tile 4]   V[slice][ci 32][tile 4]      (tile pairs XOR-swizzled by channel bit 4: conflict-free)
// Roles (fp32 MFMA and VALU do not overlap on a SIMD -- tools/mfma_valu.hip -- so every SIMD gets one wave of each
// kind: waves w and w + 4 share a SIMD):
//   waves 0-3: one (co, tile) pair per thread: dY tile straight from global registers -> A4 dY A4^T (40 packed-fp32
//              instructions) -> 36 ds_write_b32; they also move the raw x strip global -> registers -> LDS;
//   waves 4-5: one (ci, tile) window per thread, transformed rows 0-2 of B4^T d B4 (42 packed instructions);
//   waves 6-7: the same windows, transformed rows 3-5.
// Everything is double buffered (dM, V, raw strip: 144 KB of LDS), so a chunk needs ONE barrier: while the MFMAs of
// chunk c read dM/V[c&1], chunk c+1 is transformed into dM/V[(c+1)&1] from strip[(c+1)&1] / registers and the raw
// rows of chunk c+2 go into strip[c&1].
// Epilogue: every wave applies G4^T . G4 to its own 3x3 block of slices in registers, the four partial 3x3 results of a
// (co, ci) pair are summed through LDS, and the K slice's partial dW goes to a slab [tap 9][CoutP][CinQ] -- a quarter
// of the 36-slice dU.  wino44_reduce_kernel sums the slabs in a fixed order (no float atomics) and writes dW (one
// follow-up launch instead of the F(2x2) kernel's two).
#include "common.h"
#include "wgrad_reduce.h"
#include <cstring>

namespace {
inline int rup44(int v, int m) { return (v + m - 1) / m * m; }

constexpr int NS44 = 36;                // Winograd slices
constexpr int GT44 = 4;                 // tiles per chunk
constexpr int CI44 = 32;                // input channels per workgroup
constexpr int CO44 = 64;                // output channels per workgroup

struct W44Args {
    const float* x;
    const float* dy;
    float* ws;                          // [slab][tap 9][CoutP][CinQ]: per-K-slice partial dW
    float* bsum;                        // [slab][CoutP] per-slice sums of dY per output channel (bias gradient), or null
    int S, Cin, Cout, CoutP, CinQ;
    int nchunks, chunks_per_slice;
};

__device__ __forceinline__ f32x2 pk_nmul2_add(f32x2 a, f32x2 c) {        // c - 2 a
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(a), "v"(c));
    return d;
}
__device__ __forceinline__ f32x2 pk_lo_pm_2hi(f32x2 p) {                 // [p.x + 2 p.y, p.x - 2 p.y]
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, 2.0, %1 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(p));
    return d;
}
__device__ __forceinline__ f32x2 pk_xlo_pm_2yhi(f32x2 e, f32x2 f) {      // [e.x + 2 f.y, e.x - 2 f.y]
    f32x2 d;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]" : "=v"(d) : "v"(f), "v"(e));
    return d;
}

#ifdef VF_STAMPS44
// diagnostic build only: shader-clock sums per role (0 = dM waves, 1 = V waves): [role][0] chunk start -> barrier
// arrival, [1] inside the barrier, [2] barrier -> chunk end, [3] chunks, [4] prologue, [5] epilogue, [6] waves
__device__ unsigned long long g_stamps44[2][16];     // [8..13]: epilogue phases
#define VF_ST44(V) const unsigned long long V = __builtin_amdgcn_s_memtime()
#else
#define VF_ST44(V)
#endif

template <int LOGW, int MODE>
__global__ __launch_bounds__(512, 2) void wino44_wgrad_kernel(W44Args a) {
    constexpr int W = 1 << LOGW, H = W;
    constexpr int SW = MODE == 2 ? W / 2 : W, SH = SW;      // stored input size (MODE 2: nearest-upsampled x2 on read)
    constexpr int TPR = W / 4 < GT44 ? W / 4 : GT44;        // tiles of one tile row inside a chunk (4; 2 on 8x8 maps)
    constexpr int TR = GT44 / TPR;                          // tile rows per chunk (1; 2 on 8x8 maps)
    constexpr int NR = 4 * TR + 2;                          // input rows of the raw strip (6 / 10)
    constexpr int PXW = 4 * TPR;                            // pixels per strip row (16 / 8)
    constexpr int QPR = PXW / 4;                            // float4 per strip row
    // strip row: idx 4 = left halo (pixel -1), 5 .. 4+PXW = pixels, 5+PXW = right halo: a window (6 pixels from idx
    // 4 t4 + 4) is one aligned ds_read_b128 + one ds_read_b64.  Row stride 24 (20 on 8x8 maps), channel stride = 4 units
    // of 16 bytes mod 16: the b128 reads of a lane group (4 channels x 4 tiles) tile the 64 banks.
    constexpr int GXW = TR == 1 ? 24 : 20;
    constexpr bool HALO = W > PXW;                          // strip narrower than the map: halo columns carry data
    constexpr int NX4 = CI44 * NR * QPR;                    // float4 of a strip (768 / 640): <= 3 per staging thread
    constexpr int NHS = HALO ? CI44 * NR * 2 : 0;           // halo scalars (384): <= 2 per staging thread
    constexpr int CPR = (W / 4) / TPR;                      // chunks per (group of TR) tile rows
    constexpr int CPI = (H / 4) / TR * CPR;                 // chunks per image
    constexpr int MSZ = NS44 * CO44 * GT44;                 // floats of one dM buffer
    constexpr int VSZ = NS44 * CI44 * GT44;                 // floats of one V buffer
    constexpr int XCS = NR * GXW;                           // 144 (= 4 mod 16 units of 16 bytes: conflict-free) / 200 (8x8 maps: 2-way)
    constexpr int XSZ = CI44 * XCS;
    static_assert((2 * MSZ + 2 * VSZ + 2 * XSZ) * 4 <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(16))) float lds[2 * MSZ + 2 * VSZ + 2 * XSZ];
    float* const Ml = lds;                                  // dM[2][slice][co][tile]
    float* const Vl = lds + 2 * MSZ;                        // V [2][slice][ci][tile]
    float* const Xl = lds + 2 * MSZ + 2 * VSZ;              // raw x strip [2][ci][NR rows][GXW]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform (SGPR): roles branch on it
#ifdef VF_STAMPS44
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_n = 0;
    VF_ST44(t_k0);
    const unsigned long long rt_k0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int cw = wid & 1, sg = wid >> 1;
    const int li = lane & 31, lh = lane >> 5;
    const bool is_dm = wid < 4;                             // wave-uniform role
    const int vhalf = (wid >> 1) & 1;                       // V waves: 0 = transformed rows 0-2 (waves 4,5), 1 = rows 3-5
    // XCD-aware decode of (co tile, ci tile, K slice), as in the F(2x2) kernel: the workgroups of ONE K slice stream the
    // same x and dY chunks, consecutive logical ids run on the same XCD.
    const unsigned lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned lgc = xcd_remap(lin, gridDim.x * gridDim.y * gridDim.z);
    const int bx = lgc % gridDim.x, by = (lgc / gridDim.x) % gridDim.y, bz = lgc / (gridDim.x * gridDim.y);
    const int co0 = bx * CO44, ci0 = by * CI44;
    const int c_begin = bz * a.chunks_per_slice;
    const int c_end = min(a.nchunks, c_begin + a.chunks_per_slice);
    const int n = c_end - c_begin;
    const int clast = c_end - 1;

    typedef unsigned long long mask_t;
#define VF_LANES(M) __builtin_amdgcn_inverse_ballot_w64(M)
#define VF_G1 __attribute__((address_space(1)))

    // ---- dM role (tid < 256): (co, tile) = (tid >> 2, tid & 3).  Channels beyond Cout / Cin are CLAMPED, not masked:
    // they load valid (duplicate) data whose products land in rows / columns of dU that nobody reads.
    const int tch = (tid & 255) >> 2, tt = tid & 3;
    const int ttr = tt / TPR, ttc = tt % TPR;
    const int tslot = (((tt >> 1) ^ ((tch >> 4) & 1)) << 1) | (tt & 1);          // swizzled tile slot
    unsigned dyoffb = 4u * (unsigned)((min(co0 + tch, a.Cout - 1) - co0) * (H * W) + 4 * ttr * W + 4 * ttc);
    // raw x staging duty of the same waves.  A wave-slot (wave w, slot i) moves ONE strip row of 16 channels (TR = 1: 64
    // lanes = 16 ci x 4 float4) or of all 32 channels (8x8 maps: 32 ci x 2 float4): c = 4 i + w -> (row, channel half).
    // The row is wave-uniform, so "this row lies outside the image" is a scalar branch -- no lane masks, no mask
    // arithmetic in the chunk loop.  Halo scalars: slot c = 4 i + w < NR holds row c: 64 lanes = 32 ci x {left, right}.
    constexpr int NXS = TR == 1 ? 12 : 10;                 // row slots of a strip
    int xrow[3];                                           // (uniform) strip row of slot i, or -1: no duty
    unsigned xoffb[3];
    int xsl[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int c = 4 * i + (wid & 3);
        const int row = c % NR, chalf = c / NR;
        const int ci = TR == 1 ? chalf * 16 + (lane >> 2) : (lane >> 1), q = TR == 1 ? (lane & 3) : (lane & 1);
        const int cic = min(ci0 + ci, a.Cin - 1) - ci0;
        xrow[i] = c < NXS ? row : -1;
        // source row gy = 4p - 1 + row (MODE 2: stored row (gy >> 1) = 2p - 1 + ((row + 1) >> 1)); the chunk-dependent
        // part, including the "- 1", lives in the uniform base
        xoffb[i] = 4u * (unsigned)(MODE == 2 ? cic * (SH * SW) + ((row + 1) >> 1) * SW + 2 * q : cic * (H * W) + row * W + 4 * q);
        xsl[i] = ci * XCS + row * GXW + 5 + 4 * q;
    }
    int hrow[2];
    unsigned hoffb[2];
    int hsl[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = 4 * i + (wid & 3);
        const int hci = lane >> 1, hside = lane & 1;
        const int cic = min(ci0 + hci, a.Cin - 1) - ci0;
        hrow[i] = HALO && c < NR ? c : -1;
        // halo pixel gx = 4 q0 - 1 (left) or 4 q0 + PXW (right); the "4 q0 - 1" is in the base
        hoffb[i] = 4u * (unsigned)(MODE == 2 ? cic * (SH * SW) + ((c + 1) >> 1) * SW + (hside ? PXW / 2 + 1 : 0)
                                             : cic * (H * W) + c * W + (hside ? PXW + 1 : 0));
        hsl[i] = hci * XCS + c * GXW + (hside ? 5 + PXW : 4);
    }
    constexpr mask_t EVEN_LANES = 0x5555555555555555ull, ODD_LANES = 0xAAAAAAAAAAAAAAAAull;   // left / right halo lanes

    // ---- V role (tid >= 256): 128 windows x 2 halves; wave (4 + 2 vhalf + j): windows (ci, tile) = (16 j + lane >> 2, lane & 3)
    const int vci = ((wid & 1) << 4) + (lane >> 2), vtt = lane & 3;
    const int vtr = vtt / TPR, vtc = vtt % TPR;
    const int vslot = (((vtt >> 1) ^ ((vci >> 4) & 1)) << 1) | (vtt & 1);
    const int wpo = vci * XCS + (4 * vtr + vhalf) * GXW + 4 * vtc + 4;           // window origin: first row this half reads
    const int vwo = vci * GT44 + vslot + (18 * vhalf) * (CI44 * GT44);           // V slot of this half's first slice

    // raw x rows and dY tiles in flight, one chunk deep (a second register set for either, i.e. two chunks of lead,
    // was measured: no gain -- the loads are not late).  (Named registers + macros: arrays behind lambdas end up in
    // scratch memory.)
    const f32x4 z4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xr0A = z4, xr1A = z4, xr2A = z4;
    float xh0A = 0.f, xh1A = 0.f;
    f32x4 dy0 = z4, dy1 = z4, dy2 = z4, dy3 = z4;

    struct Chunk {                                     // uniform: position of a chunk and its edge flags
        const char* xb;                                // base of the x loads (one row above / one pixel left of the strip)
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
        const long long xo = MODE == 2 ? img + (long long)(2 * p - 1) * SW + 2 * q0 : img + (long long)(4 * p - 1) * W + 4 * q0;
        k.xb = uniform_ptr(reinterpret_cast<const char*>(a.x) + 4 * xo);
        k.db = uniform_ptr(reinterpret_cast<const char*>(a.dy) + 4 * ((((long long)s * a.Cout + co0) * H + 4 * p) * W + 4 * q0));
        k.top = p == 0;
        k.bot = p + TR == H / 4;
        k.left = q0 == 0;
        k.right = q0 + TPR == W / 4;
        return k;
    };
    // (the opaque copies keep base and offset apart until instruction selection: SGPR base + zero-extended VGPR offset)
#define VF_ROW_OUT(K, ROW) (((K).top && (ROW) == 0) || ((K).bot && (ROW) == NR - 1))      /* scalar */
    // Every load of the chunk loop is issued UNCONDITIONALLY (round 4): a load inside a branch costs the compiler its
    // count of the requests in flight, and every later s_waitcnt then also waits for the youngest ones (measured on the
    // F(4x4) forward kernel: 670 cycles per chunk).  A strip row outside the image is read from its neighbour row inside
    // (valid memory, scalar base shift) and replaced by zeros at the store; a halo pixel beyond the left / right image
    // border is read one pixel further in (per-lane offset select) and zeroed at the store.
#define VF_XLOAD1(K, I, R)                                                                                \
    if (xrow[I] >= 0) {                                                                                   \
        const char* b_ = (K).xb;                                                                          \
        if ((K).top && xrow[I] == 0) b_ += (MODE == 2 ? 4 * SW : 4 * W);                                  \
        if ((K).bot && xrow[I] == NR - 1) b_ -= (MODE == 2 ? 4 * SW : 4 * W);                             \
        unsigned& o_ = xoffb[I];                                                                          \
        asm("" : "+s"(b_), "+v"(o_));                                                                     \
        if (MODE == 2) {                                                                                  \
            const f32x2 h_ = *(const VF_G1 f32x2*)((const VF_G1 char*)b_ + o_);                           \
            R = (f32x4){h_.x, h_.x, h_.y, h_.y};                                                          \
        } else {                                                                                          \
            R = *(const VF_G1 f32x4*)((const VF_G1 char*)b_ + o_);                                        \
        }                                                                                                 \
    }
#define VF_HLOAD1(K, I, R)                                                                                \
    if (HALO && hrow[I] >= 0) {                                                                           \
        const char* b_ = (K).xb - 4;                                                                      \
        if ((K).top && hrow[I] == 0) b_ += (MODE == 2 ? 4 * SW : 4 * W);                                  \
        if ((K).bot && hrow[I] == NR - 1) b_ -= (MODE == 2 ? 4 * SW : 4 * W);                             \
        /* lanes whose pixel lies beyond the image border read their neighbour inside (+- one pixel) */    \
        unsigned o_ = hoffb[I];                                                                           \
        if ((K).left) o_ += VF_LANES(EVEN_LANES) ? 4u : 0u;                                               \
        if ((K).right) o_ -= VF_LANES(ODD_LANES) ? 4u : 0u;                                               \
        asm("" : "+s"(b_), "+v"(o_));                                                                     \
        R = *(const VF_G1 float*)((const VF_G1 char*)b_ + o_);                                            \
    }
#define VF_LOAD_X(K, SET)                                                                                 \
    {                                                                                                     \
        VF_XLOAD1(K, 0, xr0##SET);                                                                        \
        VF_XLOAD1(K, 1, xr1##SET);                                                                        \
        VF_XLOAD1(K, 2, xr2##SET);                                                                        \
        VF_HLOAD1(K, 0, xh0##SET);                                                                        \
        VF_HLOAD1(K, 1, xh1##SET);                                                                        \
    }
    // strip pixels sit at odd dword offsets (idx 5 + 4q): four ds_write_b32 (as two ds_write2_b32) per float4 -- same LDS
    // cost as one ds_write_b128 (MI355X_MICROARCH.md, LDS table)
#define VF_XSTORE1(K, BUF, I, R)                                                                          \
    if (xrow[I] >= 0) {                                                                                   \
        float* d_ = Xl + (BUF) * XSZ + xsl[I];                                                            \
        if (VF_ROW_OUT(K, xrow[I])) { d_[0] = 0.f; d_[1] = 0.f; d_[2] = 0.f; d_[3] = 0.f; }               \
        else { d_[0] = R.x; d_[1] = R.y; d_[2] = R.z; d_[3] = R.w; }                                      \
    }
#define VF_HSTORE1(K, BUF, I, R)                                                                          \
    if (HALO && hrow[I] >= 0) {                                                                           \
        float* d_ = Xl + (BUF) * XSZ + hsl[I];                                                            \
        const mask_t zm_ = VF_ROW_OUT(K, hrow[I]) ? ~0ull : ((K).left ? EVEN_LANES : ((K).right ? ODD_LANES : 0ull)); \
        d_[0] = VF_LANES(zm_) ? 0.f : R;                                                                  \
    }
#define VF_STORE_X(K, BUF, SET)                       /* K: the chunk the registers hold */                  \
    {                                                                                                     \
        VF_XSTORE1(K, BUF, 0, xr0##SET);                                                                  \
        VF_XSTORE1(K, BUF, 1, xr1##SET);                                                                  \
        VF_XSTORE1(K, BUF, 2, xr2##SET);                                                                  \
        VF_HSTORE1(K, BUF, 0, xh0##SET);                                                                  \
        VF_HSTORE1(K, BUF, 1, xh1##SET);                                                                  \
    }
    auto load_dy = [&](const Chunk& k) {
        {
            const char* b = k.db;
            unsigned& o = dyoffb;
            asm("" : "+s"(b), "+v"(o));
            dy0 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o);
            dy1 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o + 4 * W);
            dy2 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o + 8 * W);
            dy3 = *(const VF_G1 f32x4*)((const VF_G1 char*)b + o + 12 * W);
        }
    };

    // ---- dM = A4 dY A4^T,  A4 (6x4) = [[1,0,0,0],[1,1,1,1],[1,-1,1,-1],[1,2,4,8],[1,-2,4,-8],[0,0,0,1]]:
    //   1-D:  m0 = r0,  m1/m2 = (r0 + r2) +- (r1 + r3),  m3/m4 = (r0 + 4 r2) +- 2 (r1 + 4 r3),  m5 = r3
    // vertical pass on the two column pairs (16 packed instructions), then per transformed row (c0,c1),(c2,c3):
    //   (e, o) = (c0,c1) + (c2,c3) -> [e + o, e - o];  (e', t) = (c0,c1) + 4 (c2,c3) -> [e' + 2t, e' - 2t]   (4 each)
    float bias1 = 0.f;                                 // sum of this thread's dY tiles = dM[1][1] (bias gradient rides along)
    // Row groups {0,1,2}, {3,4}, {5} are evaluated straight from the dY registers (the vertical combinations are
    // recomputed per group instead of kept: 16 live registers less), each row then takes 4 packed instructions + 6 LDS
    // values.
    auto dy_out = [&](int buf, int i, f32x2 pa, f32x2 pb, bool count) {      // transformed row i: (c0,c1) = pa, (c2,c3) = pb
        float* mo = Ml + buf * MSZ + tch * GT44 + tslot;
        const f32x2 eo = pk_add(pa, pb);
        const f32x2 m12 = pk_lo_pm_hi(eo, eo);
        const f32x2 m34 = pk_lo_pm_2hi(pk_fmak<4>(pb, pa));
        if (i == 1 && count) bias1 += m12.x;
        mo[(6 * i + 0) * (CO44 * GT44)] = pa.x;
        mo[(6 * i + 1) * (CO44 * GT44)] = m12.x;
        mo[(6 * i + 2) * (CO44 * GT44)] = m12.y;
        mo[(6 * i + 3) * (CO44 * GT44)] = m34.x;
        mo[(6 * i + 4) * (CO44 * GT44)] = m34.y;
        mo[(6 * i + 5) * (CO44 * GT44)] = pb.y;
    };
    auto dy_group = [&](int buf, int grp, bool count) {
        if (grp == 0) {                                // rows 0, 1, 2:  r0,  (r0 + r2) +- (r1 + r3)
            const f32x2 eA = pk_add(dy0.xy, dy2.xy), oA = pk_add(dy1.xy, dy3.xy);
            const f32x2 eB = pk_add(dy0.zw, dy2.zw), oB = pk_add(dy1.zw, dy3.zw);
            dy_out(buf, 0, dy0.xy, dy0.zw, false);
            dy_out(buf, 1, pk_add(eA, oA), pk_add(eB, oB), count);
            dy_out(buf, 2, pk_sub(eA, oA), pk_sub(eB, oB), false);
        } else if (grp == 1) {                         // rows 3, 4:  (r0 + 4 r2) +- 2 (r1 + 4 r3)
            const f32x2 eA = pk_fmak<4>(dy2.xy, dy0.xy), tA = pk_fmak<4>(dy3.xy, dy1.xy);
            const f32x2 eB = pk_fmak<4>(dy2.zw, dy0.zw), tB = pk_fmak<4>(dy3.zw, dy1.zw);
            dy_out(buf, 3, pk_fmak<2>(tA, eA), pk_fmak<2>(tB, eB), false);
            dy_out(buf, 4, pk_nmul2_add(tA, eA), pk_nmul2_add(tB, eB), false);
        } else {                                       // row 5:  r3
            dy_out(buf, 5, dy3.xy, dy3.zw, false);
        }
    };

    // ---- V = B4^T d B4,  B4^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
    // window rows d0..d5 as pairs (p0,p1),(p2,p3),(p4,p5).  Vertical pass, rows 0-2 (half 0):
    //   P = d4 - 4 d2;  t0 = 4 d0 + (P - d2);  R = d3 - 4 d1;  t1 = P + R;  t2 = P - R
    // rows 3-5 (half 1):  s = d3 - d1;  q = d4 - d2;  t3 = q + 2 s;  t4 = q - 2 s;  t5 = (d5 - d3) - 4 s
    // (6 packed instructions per column pair = 18), then per transformed row the same 1-D transform along the columns
    // (8 packed instructions, below).
    f32x2 wr[5][3];                                    // the five window rows this half needs: half 0 d0..d4, half 1 d1..d5
    auto win_read = [&](int buf, int j) {              // row j of this half
        const float* p = Xl + buf * XSZ + wpo + j * GXW;
        const f32x4 q = *reinterpret_cast<const f32x4*>(p);
        const f32x2 r = *reinterpret_cast<const f32x2*>(p + 4);
        wr[j][0] = q.xy; wr[j][1] = q.zw; wr[j][2] = r;
    };
    f32x2 tv[3][3];                                    // the three transformed rows of this half
    auto win_rows = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (vhalf == 0) {                          // wr[j] = d_j
                const f32x2 P = pk_nmul4_add(wr[2][c], wr[4][c]);
                const f32x2 R = pk_nmul4_add(wr[1][c], wr[3][c]);
                tv[0][c] = pk_fmak<4>(wr[0][c], pk_sub(P, wr[2][c]));
                tv[1][c] = pk_add(P, R);
                tv[2][c] = pk_sub(P, R);
            } else {                                   // wr[j] = d_{j+1}
                const f32x2 s = pk_sub(wr[2][c], wr[0][c]);
                const f32x2 q = pk_sub(wr[3][c], wr[1][c]);
                tv[0][c] = pk_fmak<2>(s, q);
                tv[1][c] = pk_nmul2_add(s, q);
                tv[2][c] = pk_nmul4_add(s, pk_sub(wr[4][c], wr[2][c]));
            }
        }
    };
    // columns of one transformed row (p0,p1),(p2,p3),(p4,p5) -> c0..c5:
    //   P2 = (p4,p5) - 4 (p2,p3);  [c0, c5] = 4 (p0,p1) + (P2 - (p2,p3));  Q = (p2,p3) - 4 (p0,p1);  [c1, c2] = P2.x +- Q.y
    //   E = (p4,p5) - (p2,p3);  F = (p2,p3) - (p0,p1);  [c3, c4] = E.x +- 2 F.y
    auto win_col_write = [&](int buf, int r) {
        float* vo = Vl + buf * VSZ + vwo + (6 * r) * (CI44 * GT44);
        const f32x2 p01 = tv[r][0], p23 = tv[r][1], p45 = tv[r][2];
        const f32x2 P2 = pk_nmul4_add(p23, p45);
        const f32x2 c05 = pk_fmak<4>(p01, pk_sub(P2, p23));
        const f32x2 Q = pk_nmul4_add(p01, p23);
        const f32x2 c12 = pk_lo_pm_hi(P2, Q);
        const f32x2 E = pk_sub(p45, p23), F = pk_sub(p23, p01);
        const f32x2 c34 = pk_xlo_pm_2yhi(E, F);
        vo[0 * (CI44 * GT44)] = c05.x;
        vo[1 * (CI44 * GT44)] = c12.x;
        vo[2 * (CI44 * GT44)] = c12.y;
        vo[3 * (CI44 * GT44)] = c34.x;
        vo[4 * (CI44 * GT44)] = c34.y;
        vo[5 * (CI44 * GT44)] = c05.y;
    };

    f32x16 acc[9];

    // ---- prologue: chunk c0 transformed into buffers 0, strip(c0+1) in strip buffer 1, x(c0+2) and dY(c0+1) in registers
    for (int e = tid; e < 2 * XSZ; e += 512) Xl[e] = 0.f;      // channels beyond Cin / unused halo columns stay zero
    if (n > 0) {
        const Chunk k0 = chunk_at(c_begin), k1 = chunk_at(min(c_begin + 1, clast)), k2 = chunk_at(min(c_begin + 2, clast));
        if (is_dm) { VF_LOAD_X(k0, A); load_dy(k0); }
        __syncthreads();                               // zero fill done
        if (is_dm) { VF_STORE_X(k0, 0, A); VF_LOAD_X(k1, A); }
        __syncthreads();
        if (is_dm) {
            dy_group(0, 0, true);
            dy_group(0, 1, false);
            dy_group(0, 2, false);
            load_dy(k1);
            VF_STORE_X(k1, 1, A);
            VF_LOAD_X(k2, A);
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) win_read(0, j);
            win_rows();
#pragma unroll
            for (int r = 0; r < 3; ++r) win_col_write(0, r);
        }
        __syncthreads();
    }

    const int fsw = 2 * (lh ^ ((li >> 4) & 1));
    // wave (cw, sg): sg = 2 ih + jh owns the 3x3 block of slices (i, j) = (3 ih + a, 3 jh + b), local index k = 3 a + b:
    // the epilogue can then apply G4^T . G4 to its own block (sum over i in the block, j in the block) in registers
    const int ih = sg >> 1, jh = sg & 1;
    const int aoff = (18 * ih + 3 * jh) * (CO44 * GT44) + (cw * 32 + li) * GT44 + fsw;
    const int boff = (18 * ih + 3 * jh) * (CI44 * GT44) + li * GT44 + fsw;
#define VF_SL(K_) (6 * ((K_) / 3) + (K_) % 3)            /* slice offset of local index k inside the block */
    // One chunk, PAR = its buffer parity (compile time).  Side work of chunk C+1 sits BEHIND the MFMAs of slices 0-6
    // (the two waves of a SIMD run in phase: side work in front of the MFMAs would idle the matrix pipe in both at
    // once); the chunk's ONE barrier sits behind slice 6 with the fragments of slices 7 and 8 already in registers, and
    // slice 8 fetches the first fragments of chunk C+1 -- no wave waits for LDS at a chunk boundary.  The loads of the
    // chunks after that go out behind the barrier.  Chunk indices beyond the slice are clamped: the last iterations redo
    // harmless loads / LDS writes that nobody reads.
#define VF_MF(K, AV, BV, FIRST)                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    acc[K] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, (FIRST) ? (f32x16){0} : acc[K], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);
#define VF_FRAG(AD, BD, PAR_, K_)                                                                         \
    {                                                                                                     \
        AD = *reinterpret_cast<const f32x2*>(Ml + (PAR_) * MSZ + aoff + VF_SL(K_) * (CO44 * GT44));       \
        BD = *reinterpret_cast<const f32x2*>(Vl + (PAR_) * VSZ + boff + VF_SL(K_) * (CI44 * GT44));       \
    }
#ifdef VF_STAMPS44
#define VF_ST44_ACC1() { st_a += t_c1 - t_c0; st_b += t_c2 - t_c1; t_mid = t_c2; }
#define VF_ST44_ACC2() { const unsigned long long t_c3 = __builtin_amdgcn_s_memtime(); st_c += t_c3 - t_mid; st_n += 1; }
    unsigned long long t_mid = 0;
#else
#define VF_ST44_ACC1()
#define VF_ST44_ACC2()
#endif
#define VF_W44_CHUNK(C, PAR, FIRST, SIDE, SP)                                                             \
    {                                                                                                     \
        const Chunk kn2 = chunk_at(min((C) + 2, clast)), kn4 = chunk_at(min((C) + 3, clast));             \
        f32x2 a2, b2;                 /* fragments run TWO slices ahead of their MFMAs: slice k fetches slice k+2 */ \
        VF_ST44(t_c0);                                                                                    \
        _Pragma("unroll") for (int k = 0; k < 9; ++k) {                                                   \
            VF_MF(k, a_cur.x, b_cur.x, FIRST);                                                            \
            if (k < 7) { VF_FRAG(a2, b2, PAR, k + 2); }                                                   \
            else { VF_FRAG(a2, b2, (PAR) ^ 1, k - 7); }   /* behind the barrier: chunk C+1, slices 0 and 1 */ \
            SIDE(C, PAR, k, 0, kn2, kn4, SP);                                                               \
            VF_MF(k, a_cur.y, b_cur.y, false);                                                            \
            SIDE(C, PAR, k, 1, kn2, kn4, SP);                                                               \
            if (k == 6) { VF_ST44(t_c1); __syncthreads(); VF_ST44(t_c2); VF_ST44_ACC1(); }              \
            a_cur = a_nx; b_cur = b_nx; a_nx = a2; b_nx = b2;                                             \
        }                                                                                                 \
        VF_ST44_ACC2();                                                                                   \
    }
    // dM waves: raw rows of chunk C+2 -> strip[PAR] in slice 0, transform of dY(C+1) spread over slices 1-4; every
    // global load goes out as soon as its registers are free (x: chunk C+3, dY: chunk C+2)
#define VF_SIDE_DM(C, PAR, K, J, KN2, KN4, SP)                                                            \
    {                                                                                                     \
        if ((K) == 0 && (J) == 0) VF_STORE_X(KN2, PAR, SP);                                               \
        if ((K) == 0 && (J) == 1) VF_LOAD_X(KN4, SP);      /* chunk C+3, into the registers just freed */ \
        if ((K) == 1 && (J) == 0) dy_group((PAR) ^ 1, 0, (C) + 1 <= clast);                               \
        if ((K) == 2 && (J) == 0) dy_group((PAR) ^ 1, 1, false);                                          \
        if ((K) == 3 && (J) == 0) { dy_group((PAR) ^ 1, 2, false); load_dy(KN2); }                        \
    }
#define VF_SIDE_V(C, PAR, K, J, KN2, KN4, SP)                                                                  \
    {                                                                                                     \
        if ((K) == 0 && (J) == 1) { win_read((PAR) ^ 1, 0); win_read((PAR) ^ 1, 1); win_read((PAR) ^ 1, 2); } \
        if ((K) == 1 && (J) == 0) { win_read((PAR) ^ 1, 3); win_read((PAR) ^ 1, 4); }                     \
        if ((K) == 2 && (J) == 0) win_rows();                                                             \
        if ((K) == 3 && (J) == 0) win_col_write((PAR) ^ 1, 0);                                            \
        if ((K) == 4 && (J) == 0) win_col_write((PAR) ^ 1, 1);                                            \
        if ((K) == 5 && (J) == 0) win_col_write((PAR) ^ 1, 2);                                            \
    }
#ifdef VF_STAMPS44
    VF_ST44(t_k1);
#endif
    f32x2 a_cur, b_cur, a_nx, b_nx;
    VF_FRAG(a_cur, b_cur, 0, 0);
    VF_FRAG(a_nx, b_nx, 0, 1);
    if (n <= 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = (f32x16){0};
    } else if (is_dm) {
        VF_W44_CHUNK(c_begin, 0, true, VF_SIDE_DM, A);
        int c = c_begin + 1;
        for (; c + 1 < c_end; c += 2) {
            VF_W44_CHUNK(c, 1, false, VF_SIDE_DM, A);
            VF_W44_CHUNK(c + 1, 0, false, VF_SIDE_DM, A);
        }
        if (c < c_end) VF_W44_CHUNK(c, 1, false, VF_SIDE_DM, A);
    } else {
        VF_W44_CHUNK(c_begin, 0, true, VF_SIDE_V, A);
        int c = c_begin + 1;
        for (; c + 1 < c_end; c += 2) {
            VF_W44_CHUNK(c, 1, false, VF_SIDE_V, A);
            VF_W44_CHUNK(c + 1, 0, false, VF_SIDE_V, A);
        }
        if (c < c_end) VF_W44_CHUNK(c, 1, false, VF_SIDE_V, A);
    }
#undef VF_W44_CHUNK
#undef VF_SIDE_DM
#undef VF_SIDE_V
#undef VF_FRAG
#undef VF_MF
#undef VF_XLOAD1
#undef VF_LOAD_X
#undef VF_STORE_X
#undef VF_HLOAD1
#undef VF_XSTORE1
#undef VF_HSTORE1
#undef VF_G1
#undef VF_LANES

#undef VF_SL
#ifdef VF_STAMPS44
    VF_ST44(t_k2);
#endif
    const int slab = bz;
    if (a.bsum && by == 0 && is_dm) {          // bias gradient partial: the 4 tile lanes of a channel, fixed order
        float b = bias1;
        b += __shfl_xor(b, 1, 64);
        b += __shfl_xor(b, 2, 64);
        if (tt == 0) a.bsum[(size_t)slab * a.CoutP + co0 + tch] = b;
    }

    // ---- epilogue: partial dW = G4^T dU G4 of this K slice, 9 values per (co, ci) instead of 36.
    //   G4 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    // 1-D, lower half (rows 0-2 of G4): y0 = u0/4 - (u1+u2)/6,  y1 = (u2-u1)/6,  y2 = -(u1+u2)/6
    //      upper half (rows 3-5):       y0 = (u0+u1)/24,        y1 = (u0-u1)/12, y2 = (u0+u1)/6 + u2
    // Every wave transforms its own 3x3 block (columns by jh, rows by ih); the four partial 3x3 results of a (co, ci)
    // pair are then summed through LDS in a fixed order: (ih=1) -> (ih=0), then (jh=1) -> (jh=0).
    auto g3 = [](float u0, float u1, float u2, int half, float& y0, float& y1, float& y2) {
        if (half == 0) {
            const float sm = u1 + u2, d = u2 - u1;
            const float s6 = sm * (-1.0f / 6.0f);
            y0 = __builtin_fmaf(0.25f, u0, s6);
            y1 = d * (1.0f / 6.0f);
            y2 = s6;
        } else {
            const float sm = u0 + u1, d = u0 - u1;
            y0 = sm * (1.0f / 24.0f);
            y1 = d * (1.0f / 12.0f);
            y2 = __builtin_fmaf(sm, 1.0f / 6.0f, u2);
        }
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float t[3][3];
#pragma unroll
        for (int a3 = 0; a3 < 3; ++a3)              // columns: sum over j of this block
            g3(acc[3 * a3][r], acc[3 * a3 + 1][r], acc[3 * a3 + 2][r], jh, t[a3][0], t[a3][1], t[a3][2]);
#pragma unroll
        for (int q = 0; q < 3; ++q) {              // rows: sum over i of this block
            float y0, y1, y2;
            g3(t[0][q], t[1][q], t[2][q], ih, y0, y1, y2);
            acc[q][r] = y0; acc[3 + q][r] = y1; acc[6 + q][r] = y2;       // acc[3 p + q] = partial dW[p][q]
        }
    }
    // cross-wave sum through LDS (the loop's images are dead: every wave is past the last chunk's barrier)
    float* const xch = lds;                        // [slot][k 9][r 16][lane 64]
    VF_ST44(t_e1);
    __syncthreads();
    VF_ST44(t_e2);
    if (ih == 1) {
        float* o = xch + (size_t)(2 * jh + cw) * (9 * 16 * 64) + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(k * 16 + r) * 64] = acc[k][r];
    }
    __syncthreads();
    VF_ST44(t_e3);
    if (ih == 0) {
        const float* o = xch + (size_t)(2 * jh + cw) * (9 * 16 * 64) + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[k][r] += o[(k * 16 + r) * 64];
    }
    VF_ST44(t_e4);
    __syncthreads();
    if (ih == 0 && jh == 1) {
        float* o = xch + (size_t)cw * (9 * 16 * 64) + lane;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(k * 16 + r) * 64] = acc[k][r];
    }
    __syncthreads();
    VF_ST44(t_e5);
    if (sg == 0) {
        const float* o = xch + (size_t)cw * (9 * 16 * 64) + lane;
        // slab [tap 9][CoutP][CinQ]: one 64-bit base per lane, 32-bit offsets inside the slab
        float* sb = a.ws + (size_t)slab * 9 * a.CoutP * a.CinQ + (size_t)(co0 + cw * 32 + 4 * lh) * a.CinQ + ci0 + li;
        const int kst = a.CoutP * a.CinQ;
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                sb[(size_t)k * kst + ((r & 3) + 8 * (r >> 2)) * a.CinQ] = acc[k][r] + o[(k * 16 + r) * 64];
    }
#ifdef VF_STAMPS44
    if (lane == 0) {
        const unsigned long long t_k3 = __builtin_amdgcn_s_memtime();
        unsigned long long* g = g_stamps44[is_dm ? 0 : 1];
        atomicAdd(&g[0], st_a); atomicAdd(&g[1], st_b); atomicAdd(&g[2], st_c); atomicAdd(&g[3], st_n);
        atomicAdd(&g[4], t_k1 - t_k0); atomicAdd(&g[5], t_k3 - t_k2); atomicAdd(&g[6], 1ull);
        atomicAdd(&g[8], t_e1 - t_k2); atomicAdd(&g[9], t_e2 - t_e1); atomicAdd(&g[10], t_e3 - t_e2);
        atomicAdd(&g[11], t_e4 - t_e3); atomicAdd(&g[12], t_e5 - t_e4); atomicAdd(&g[13], t_k3 - t_e5);
        atomicAdd(&g[14], t_k3 - t_k0); atomicAdd(&g[15], __builtin_amdgcn_s_memrealtime() - rt_k0);   // shader clocks / 100 MHz ticks
    }
#endif
}

// Slab sum: dW[co][ci][tap] = sum over the K slices' partial dW (fixed order, no float atomics).  One workgroup = one
// output channel x 64 input channels; thread (ci, g) sums all nine taps of every fourth slab, the four partial sums
// meet in LDS, and the 9 x 64 results are written out as one contiguous 2304-byte run of dW.
// Trailing workgroups: db[co] = sum over the slices' dY sums.
__device__ __forceinline__ void wino44_reduce_body(const float* __restrict__ ws, float* __restrict__ dw,
                                                   int Cout, int Cin, int CoutP, int CinQ, int nslab,
                                                   const float* __restrict__ bsum, float* __restrict__ db,
                                                   float* db2, int nmain, int bid) {
    if (bid >= nmain) {
        __shared__ float red[4][64];
        const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
        const int co = (bid - nmain) * 64 + cx;
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
    __shared__ float part[3][9][64];
    __shared__ float dwl[64 * 9];
    const int cx = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ncb = (Cin + 63) / 64;
    const int co = bid / ncb, cb = (bid % ncb) * 64, ci = cb + cx;
    const size_t kst = (size_t)CoutP * CinQ, sst = 9 * kst;
    // thread (ci, g): all nine taps of the slabs z = g, g + 4, ...; two slabs (18 independent loads) in flight
    float s9[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s9[k] = 0.f;
    if (ci < CinQ) {
        const float* p = ws + (size_t)co * CinQ + ci;
        float c0[9], c1[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) c0[k] = c1[k] = 0.f;
        int z = g;
        for (; z + 4 < nslab; z += 8) {
            float v0[9], v1[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                v0[k] = p[(size_t)z * sst + k * kst];
                v1[k] = p[(size_t)(z + 4) * sst + k * kst];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) { c0[k] += v0[k]; c1[k] += v1[k]; }
        }
        if (z < nslab) {
#pragma unroll
            for (int k = 0; k < 9; ++k) c0[k] += p[(size_t)z * sst + k * kst];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) s9[k] = c0[k] + c1[k];
    }
    if (g > 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) part[g - 1][k][cx] = s9[k];
    }
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) dwl[cx * 9 + k] = (s9[k] + part[0][k][cx]) + (part[1][k][cx] + part[2][k][cx]);
    }
    __syncthreads();
    const int nvalid = min(64, Cin - cb) * 9;
    float* o = dw + ((size_t)co * Cin + cb) * 9;
    for (int e = threadIdx.x; e < nvalid; e += 256) o[e] = dwl[e];
}

__global__ __launch_bounds__(256) void wino44_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw,
                                                            int Cout, int Cin, int CoutP, int CinQ, int nslab,
                                                            const float* __restrict__ bsum, float* __restrict__ db,
                                                            float* db2, int nmain) {
    wino44_reduce_body(ws, dw, Cout, Cin, CoutP, CinQ, nslab, bsum, db, db2, nmain, (int)blockIdx.x);
}

// The slab sums of MANY layers in one launch (round 5): the weight gradients are not needed before the optimizer step (or
// the segment's all-reduce), so a host may run only the main kernel per layer (vf_wino_wgrad_main), keep that layer's
// workspace alive and sum all of them here -- 65 follow-up launches of ~7 us per training iteration become one (or one
// per all-reduce segment) that runs at the chip's streaming rate.  Row = what wino44_reduce_kernel takes + the row's first
// workgroup in this launch (9 x 8 bytes).
// (W44Red: wgrad_reduce.h -- round 6: rows with nt != 0 are direct / 1x1 weight gradients, summed by wgrad_reduce_body)

__global__ __launch_bounds__(256) void wino44_reduce_multi_kernel(const W44Red* __restrict__ tab, int nrows) {
    __shared__ int row_s;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nrows - 1;                      // last row whose first workgroup is <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tab[mid].first <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        row_s = lo;
    }
    __syncthreads();
    const W44Red d = tab[row_s];
    if (d.nt != 0) {                                  // (uniform over the workgroup)
        wgrad_reduce_body(d.ws, d.dw, d.nslab, d.nt, d.Cout, d.Cin, d.CoutP, d.CinQ, (int)blockIdx.x - d.first);
        return;
    }
    wino44_reduce_body(d.ws, d.dw, d.Cout, d.Cin, d.CoutP, d.CinQ, d.nslab, d.bsum, d.db, d.db2, d.nmain,
                       (int)blockIdx.x - d.first);
}

// desc (host memory, 9 x int64, or NULL): when given, the slab sum is NOT launched; the row for vf_wino44_reduce_multi
// is written there (its `first` field left 0: the caller lays the rows out) and the row's workgroup count goes to *nblocks.
template <int LOGW, int MODE>
int launch_wino44_wgrad(W44Args a, float* dw, float* db, float* db2, size_t ws_floats, hipStream_t st,
                        long long* desc = nullptr, int* nblocks = nullptr) {
    constexpr int W = 1 << LOGW;
    a.nchunks = a.S * ((W / 4) * (W / 4) / GT44);
    const int nco = a.CoutP / CO44, nci = a.CinQ / CI44;
    const size_t slab_floats = (size_t)9 * a.CoutP * a.CinQ;
    int z = 256 / (nco * nci);
    if (z < 1) z = 1;
    if (z > a.nchunks) z = a.nchunks;
    if (ws_floats < slab_floats + 256 * (size_t)a.CoutP) return (int)hipErrorInvalidValue;
    const size_t zmax = (ws_floats - 256 * (size_t)a.CoutP) / slab_floats;
    if ((size_t)z > zmax) z = (int)zmax;
    a.chunks_per_slice = (a.nchunks + z - 1) / z;
    z = (a.nchunks + a.chunks_per_slice - 1) / a.chunks_per_slice;
    a.bsum = db ? a.ws + (size_t)z * slab_floats : nullptr;      // the per-slice dY sums sit behind the slabs
    hipLaunchKernelGGL((wino44_wgrad_kernel<LOGW, MODE>), dim3(nco, nci, z), dim3(512), 0, st, a);
    const int nmain = a.Cout * ((a.Cin + 63) / 64), nbias = db ? (a.Cout + 63) / 64 : 0;
    if (desc) {
        W44Red r;
        r.ws = a.ws; r.dw = dw; r.bsum = a.bsum; r.db = db; r.db2 = db2;
        r.Cout = a.Cout; r.Cin = a.Cin; r.CoutP = a.CoutP; r.CinQ = a.CinQ; r.nslab = z; r.nmain = nmain; r.first = 0; r.nt = 0;
        memcpy(desc, &r, sizeof(r));
        *nblocks = nmain + nbias;
        VF_RETURN_LAST_ERROR();
    }
    hipLaunchKernelGGL(wino44_reduce_kernel, dim3(nmain + nbias), dim3(256), 0, st, a.ws, dw, a.Cout, a.Cin, a.CoutP,
                       a.CinQ, z, a.bsum, db, db2, nmain);
    VF_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

// workspace floats for vf_wino_wgrad at this shape (slabs of transformed partial gradients + per-slice dY sums)
long vf_wino_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const long slab = 9L * rup44(Cout, CO44) * rup44(Cin, CI44);
    const int nco = rup44(Cout, CO44) / CO44, nci = rup44(Cin, CI44) / CI44;
    long z = 256 / (nco * nci);
    if (z < 1) z = 1;
    const long nchunks = (long)S * (H / 4) * (W / 4) / GT44;
    if (z > nchunks) z = nchunks;
    return z * slab + 256L * rup44(Cout, CO44);
}

int vf_wino_wgrad_supported(int H, int W, int mode) {
    return H == W && (W == 8 || W == 16 || W == 32 || W == 64) && (mode == 0 || mode == 2);
}

// dw[Cout][Cin][3][3] of a stride-1 3x3 conv (H = W = output size in {8, 16, 32, 64}; mode 2: x is stored at half
// size and nearest-upsampled on read) via Winograd F(4x4,3x3).
// db (or NULL): also the bias gradient sum_{s,p} dY[s][co][p] -- the kernel reads every dY tile anyway;
// db2 (or NULL): a second [Cout] destination for the same sums (the residual 1x1 conv shares this dY)
int vf_wino_wgrad(const float* x, const float* dy, float* dw, float* db, float* db2, float* ws, long ws_floats, int S,
                  int Cin, int Cout, int H, int W, int mode, void* stream) {
    if (S <= 0) return 0;
    if (!vf_wino_wgrad_supported(H, W, mode)) return (int)hipErrorInvalidValue;
    W44Args a;
    a.x = x; a.dy = dy; a.ws = ws; a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CoutP = rup44(Cout, CO44); a.CinQ = rup44(Cin, CI44);
    hipStream_t st = (hipStream_t)stream;
#define VF_WG(LW) \
    return mode == 2 ? launch_wino44_wgrad<LW, 2>(a, dw, db, db2, (size_t)ws_floats, st) \
                     : launch_wino44_wgrad<LW, 0>(a, dw, db, db2, (size_t)ws_floats, st)
    if (W == 8) VF_WG(3);
    if (W == 16) VF_WG(4);
    if (W == 32) VF_WG(5);
    VF_WG(6);
#undef VF_WG
}

// vf_wino_wgrad without its follow-up launch: the main kernel only; desc9 (HOST memory, 9 x int64) receives the row that
// vf_wino44_reduce_multi needs for this layer, *nblocks its workgroup count.  ws must stay untouched until that launch.
int vf_wino_wgrad_main(const float* x, const float* dy, float* dw, float* db, float* db2, float* ws, long ws_floats, int S,
                       int Cin, int Cout, int H, int W, int mode, long long* desc9, int* nblocks, void* stream) {
    if (!desc9 || !nblocks) return (int)hipErrorInvalidValue;
    *nblocks = 0;
    if (S <= 0) return 0;
    if (!vf_wino_wgrad_supported(H, W, mode)) return (int)hipErrorInvalidValue;
    W44Args a;
    a.x = x; a.dy = dy; a.ws = ws; a.S = S; a.Cin = Cin; a.Cout = Cout;
    a.CoutP = rup44(Cout, CO44); a.CinQ = rup44(Cin, CI44);
    hipStream_t st = (hipStream_t)stream;
#define VF_WG(LW) \
    return mode == 2 ? launch_wino44_wgrad<LW, 2>(a, dw, db, db2, (size_t)ws_floats, st, desc9, nblocks) \
                     : launch_wino44_wgrad<LW, 0>(a, dw, db, db2, (size_t)ws_floats, st, desc9, nblocks)
    if (W == 8) VF_WG(3);
    if (W == 16) VF_WG(4);
    if (W == 32) VF_WG(5);
    VF_WG(6);
#undef VF_WG
}

// table: DEVICE memory, nrows rows of 9 x int64 as written by vf_wino_wgrad_main, with the `first` field (int32 at byte 64
// of a row) = the sum of the preceding rows' workgroup counts; nblocks = the total.
int vf_wino44_reduce_multi(const void* table, int nrows, int nblocks, void* stream) {
    if (nrows <= 0 || nblocks <= 0) return 0;
    hipLaunchKernelGGL(wino44_reduce_multi_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream,
                       (const W44Red*)table, nrows);
    VF_RETURN_LAST_ERROR();
}

}  // extern "C"

#ifdef VF_STAMPS44
extern "C" void vf_debug_stamps44(unsigned long long* out16, int reset) {
    if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps44), 256);
    if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps44), z, 256); }
}
#endif
