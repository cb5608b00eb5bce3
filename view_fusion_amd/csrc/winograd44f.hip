// Fused Winograd F(4x4,3x3) convolution, forward and dgrad, for the stride-1 3x3 layers on the LARGE maps (64x64 and
// 32x32; reference model/unet.py:42,189,214):
//
//     Y (4x4) = A4^T [ sum_ci (G4 g G4^T) (.) (B4^T d B4) ] A4          d = 6x6 input window, 36 products per 16 outputs
//
// = 2.25 multiplies per output pixel and (ci, co) pair instead of 3 for the nested F(2,3)xF(4,3) kernel of
// winograd24.hip (which stays the kernel of the 16x16 / 8x8 maps, where a 512-pixel workgroup tile cannot fill the
// chip) and 9 for the direct form: 25 % fewer MFMAs for the same side work per MFMA.  fp32 throughout.  Per-layer error
// against an fp64 convolution (random data, K = 576 ... 2880): rel-L2 1.7-3.7e-6 (nested: 0.7-1.4e-6, direct fp32:
// 0.2-0.3e-6) -- the 4-point constants (4, 5, 8, 1/6, 1/24) enter on both axes; the weight transform is evaluated in
// double and rounded once.  tools/parity_margin.py reports what that does to the full UNet.
//
// Structure (one kernel, nothing transformed ever touches HBM):
//   * workgroup = 8 waves = 64 co x 32 tiles (4x4 outputs each: 512 pixels) x 36 slices, ONE per CU, persistent over the
//     whole tiles of the launch; wave (cw, g) owns 32 co x 32 tiles x 9 slices = 144 accumulators
//     (v_mfma_f32_32x32x2_f32): one full transformed row of six slices + half of a row it shares with its neighbour
//         g = 0: row 0, row 1 cols 0-2    g = 1: row 2, row 1 cols 3-5    g = 2: row 3, row 4 cols 0-2    g = 3: row 5, row 4 cols 3-5
//     (slices are stored in that "wave order" p = 9 g + j everywhere: packed weights, LDS image);
//   * U = G4 g G4^T arrives pre-transformed + packed  U[co tile][chunk][p 36][co 64][ci 8]  and goes STRAIGHT from
//     global memory into the MFMA A registers (one float4 per lane and slice, reloaded in place one chunk ahead);
//   * per 8-channel chunk the raw haloed rows are staged in LDS, one ROW per wave-slot (row / channel group are scalars),
//     every load and store unconditional (rows outside the image come from a zero buffer); two threads per
//     (channel, tile) transform the 6x6 window (B4^T d B4: three transformed rows each, 42 packed-fp32 instructions; a
//     window row = one ds_read_b128 + one ds_read_b64) into V[ci & 3][p][ci >> 2][tile]; the two halves of that duty run
//     their own copy of the chunk loop;
//   * rows and V are double buffered: ONE (LDS-only) barrier per chunk; all side work sits behind MFMAs (fp32 MFMA and
//     VALU do not overlap on a SIMD, tools/mfma_valu.hip);
//   * epilogue: A4^T along the columns in registers (full row: 4 values, half row: 3 values per accumulator register),
//     exchange through LDS in FOUR passes of four accumulator registers through two buffers (publish k+1 while finishing
//     k: one barrier per phase), every wave finishes one channel row of its half per pass (A4^T along the rows, bias +
//     per-view bias + residual, float4 row stores); the next tile's first loads go out before any store, its raw rows and
//     V(0) are staged in the last two phases.
// MODE 0: plain input; MODE 2: nearest-x2-upsampled input (Upsample conv), as in conv.hip.
#include "common.h"
#include "wino_plan.h"

namespace {

constexpr int FCO = 64;       // output channels per workgroup
constexpr int FTT = 32;       // 4x4 output tiles per workgroup
constexpr int FCK = 8;        // input channels per chunk
constexpr int FNS = 36;       // Winograd slices
constexpr int F44_PERSIST = 256;

// slice (transformed row a, transformed column b) -> position p in wave order
__host__ __device__ constexpr int f44_p(int a, int b) {
    return a == 0 ? b : a == 1 ? (b < 3 ? 6 + b : 12 + b) : a == 2 ? 9 + b : a == 3 ? 18 + b
         : a == 4 ? (b < 3 ? 24 + b : 30 + b) : 27 + b;
}

struct F44Args {
    const float* x;
    const float* u;       // packed transformed weights
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    int S, Cin, Cout, CinP, CoutP;
    // tail splitting: tiles [0, nfull) are computed whole (by the persistent workgroups); workgroup
    // npers + j*tail_split + p computes the p-th K range of tile nfull + j and leaves a raw partial output in ws
    int nfull, tail_split;
    float* ws;
    int npers;
};

template <int LOGW, int MODE>
struct FGeo {
    static constexpr int W = 1 << LOGW, H = W, HW = W * H;
    static constexpr int SH = MODE == 2 ? H / 2 : H, SW = MODE == 2 ? W / 2 : W;   // source size
    static constexpr int TWC = W / 4;                // tiles per output row
    static constexpr int THR = H / 4;                // tile rows per image
    static constexpr int TR = FTT / TWC;             // tile rows of one workgroup (2 at 64x64, 4 at 32x32)
    static constexpr int WPI = THR / TR;             // workgroups per image
    static constexpr int PH = 4 * TR + 2;            // haloed patch rows
    // patch row: idx 4 = left halo (pixel -1), 5 .. 4+W = pixels, 5+W = right halo; a window (6 pixels from idx
    // 4 t + 4) is one aligned ds_read_b128 + one ds_read_b64.  Stride: the 16 lanes of a b128 lane group take 16
    // consecutive tiles = one tile row (W = 64) or two (W = 32: patch rows 4 apart, 4 PW = 32 mod 64 dwords).
    static constexpr int PW = W + 8;
    static constexpr int PS = PH * PW;
    static constexpr int Q = W / 4;
    static_assert(TWC * THR >= FTT && TR >= 1 && WPI >= 1 && TR * TWC == FTT, "F(4x4) forward kernel: 32x32 and 64x64 maps");
    static __device__ __forceinline__ int t_row(int tl) { return tl / TWC; }
    static __device__ __forceinline__ int t_col(int tl) { return tl % TWC; }
    static int groups(int S) { return S * WPI; }
};

#ifdef VF_STAMPS44F
__device__ unsigned long long g_stamps44f[16];
#endif
// rows outside the image are loaded from here (as large as the largest per-lane offset of a row load: 3 channels of a
// 64x64 map + one row; never written)
__device__ float g_f44_zero[16384];

// An LDS-only workgroup barrier: __syncthreads() also drains the wave's outstanding GLOBAL stores (vmcnt), which in the
// epilogue are the output rows just issued -- thousands of cycles; the exchange only needs the LDS traffic ordered.

// RAGGED: Cin is no multiple of 8 (stem, dgrad of the head): the last chunk's channel offsets are clamped per lane.
template <int LOGW, int MODE, bool RAGGED>
__global__ __launch_bounds__(512, 2) void wino44_conv_kernel(F44Args a) {
    using G = FGeo<LOGW, MODE>;
    constexpr int NT_ = 512;
    constexpr int UCH = FNS * FCO * FCK;                 // floats of one (co tile, chunk) block of U
    constexpr int VSZ = FNS * FCK * FTT;
    constexpr int PSZ = FCK * G::PS;
    constexpr int XCH = 2 * 8 * 7 * 4 * 64;              // epilogue exchange: two buffers of [wave 8][value 7][reg 4][lane 64]
    constexpr int LDSF = 2 * PSZ + (2 * VSZ > XCH ? 2 * VSZ : XCH);
    static_assert(LDSF * 4 <= 160 * 1024, "LDS budget");

    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    float* const Pl = lds;                               // raw rows [2][ci 8][PH][PW] (outside the exchange area)
    float* const Vl = lds + 2 * PSZ;                     // V [2][ci & 3][p 36][ci >> 2][tile 32]; epilogue: exchange

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wid & 1, wg_ = wid >> 1;              // channel half, slice group of this wave
    const int li = lane & 31, lh = lane >> 5;
    const int ncot = a.CoutP / FCO;
    const bool partial = (int)blockIdx.x >= a.npers;
    const int tail_id = partial ? (int)blockIdx.x - a.npers : 0;
    int c0 = 0, nch = a.CinP / FCK;                      // this workgroup's chunk range [c0, c0 + nch)
    if (partial) {
        const int per = (nch + a.tail_split - 1) / a.tail_split;
        c0 = (tail_id % a.tail_split) * per;
        nch = max(0, min(nch - c0, per));
    }
    const int clast = max(nch - 1, 0);

    // ---- raw-row staging, one ROW per wave-slot (as the F(4x4) weight-gradient kernel does): slot sid = wid + 8 i
    // (i < 3) moves patch row sid % PH of CPS = 64 / Q channels (lane = (channel, float4 column)), so "this row lies
    // outside the image" and every row / channel-group address part are SCALAR: the chunk loop carries two per-lane
    // offsets (global bytes, LDS floats) instead of per-element offset / mask registers.
    constexpr int CPS = 64 / G::Q;                       // channels per slot (4 at W = 64, 8 at W = 32)
    constexpr int NSLOT = G::PH * (FCK / CPS);           // 20 / 18
    static_assert(NSLOT > 16 && NSLOT <= 24, "three slots per wave");
    const int xcl = lane / G::Q, xq = lane % G::Q;
    const unsigned xgofs = (unsigned)(xcl * (4 * G::SH * G::SW) + (MODE == 0 ? 16 : 8) * xq);
    // LDS position of this lane's store: the ALIGNED float4 at idx 4 xq + 4 = pixels 4 xq - 1 .. 4 xq + 2 (see VF_XST1)
    const int xlofs = xcl * G::PS + 4 * xq;              // (+ 4: an immediate offset of the stores)
    // lanes that own the last float4 of a row (their .w = pixel W - 1 goes to idx W + 4 with a store of its own)
    constexpr unsigned long long XLAST = G::Q == 16 ? 0x8000800080008000ull : 0x8080808080808080ull;
    static_assert(G::Q == 16 || G::Q == 8, "row = 16 or 8 lanes");

    struct Tile {                                        // all workgroup-uniform
        const char* ubase;                               // U block of (co tile, first chunk)
        const char* xbase;                               // the tile's view, channel 0
        int r0;                                          // first output row
    };
    auto tile_pos = [&](unsigned logical, int& s_, int& r0_, int& cot_) {
        cot_ = logical % ncot;
        const int wg = logical / ncot;
        s_ = wg / G::WPI;
        r0_ = (wg % G::WPI) * 4 * G::TR;                 // first output row
    };
    auto make_tile = [&](unsigned logical) -> Tile {
        Tile t;
        int ts, cot;
        tile_pos(logical, ts, t.r0, cot);
        t.ubase = uniform_ptr(a.u + ((size_t)cot * (a.CinP / FCK) + c0) * UCH);
        t.xbase = uniform_ptr(a.x + (size_t)ts * a.Cin * (G::SH * G::SW));
        return t;
    };
    unsigned uoffb = 4u * (unsigned)(((wg_ * 9) * FCO + cw * 32 + li) * FCK + 4 * lh);
    // slot I of this wave: patch row / channel group (scalar).  Every wave issues exactly three row loads and three row
    // stores per chunk, UNCONDITIONALLY (a load or an LDS store inside a branch costs the compiler its count of what is
    // in flight: every later s_waitcnt then waits for the youngest requests too -- measured: 670 cycles per chunk):
    //   * a row outside the image is loaded from a zero-filled buffer (scalar select of the base) and stored like
    //     any other row: no per-tile zeroing of such rows either;
    //   * a wave without a third slot (sid >= NSLOT) repeats its second one (same bytes to the same place).
#define VF_SID(I) (wid + 8 * (I) < NSLOT ? wid + 8 * (I) : wid + 8 * (I) - 8)
#define VF_SROW(I) (VF_SID(I) % G::PH)
#define VF_SCG(I) (VF_SID(I) / G::PH)
#define VF_SUY(T, I) ((T).r0 + VF_SROW(I) - 1)
    // channels beyond Cin (last chunk of a Cin that is no multiple of 8) read the clamped last channel -- finite
    // values whose packed weights are zero
    auto fetch_x = [&](const Tile& t, int i, int c, f32x4& v) {
        const int cg = c0 + c;
        const int uy = VF_SUY(t, i);
        const char* b_ = t.xbase + (size_t)(cg * FCK + VF_SCG(i) * CPS) * (4 * G::SH * G::SW)
                         + 4 * (MODE == 0 ? uy * G::SW : (uy >> 1) * G::SW);
        unsigned o_ = xgofs;
        if (RAGGED) {                                    // (every chunk: in all but the last one lim >= CPS - 1)
            int lim = a.Cin - 1 - cg * FCK - VF_SCG(i) * CPS;                 // last valid channel of this slot
            if (lim < 0) { b_ += (long)lim * (4 * G::SH * G::SW); lim = 0; }  // none: every lane reads channel Cin - 1
            o_ = (unsigned)(min(xcl, lim) * (4 * G::SH * G::SW) + (MODE == 0 ? 16 : 8) * xq);
        }
        if (uy < 0 || uy >= G::H) b_ = reinterpret_cast<const char*>(g_f44_zero);     // (scalar: s_cselect)
        asm("" : "+s"(b_), "+v"(o_));
        if (MODE == 0) v = *(const __attribute__((address_space(1))) f32x4*)((const __attribute__((address_space(1))) char*)b_ + o_);
        else {
            const f32x2 h = *(const __attribute__((address_space(1))) f32x2*)((const __attribute__((address_space(1))) char*)b_ + o_);
            v = (f32x4){h.x, h.x, h.y, h.y};
        }
    };

    unsigned lin = partial ? 0u : blockIdx.x;
    const unsigned tail_logical = (unsigned)(a.nfull + tail_id / a.tail_split);
    auto logical_of = [&](unsigned l) -> unsigned { return partial ? tail_logical : xcd_remap(l, a.nfull); };
    Tile cur = make_tile(logical_of(lin));

    f32x4 ur0, ur1, ur2, ur3, ur4, ur5, ur6, ur7, ur8;   // U fragments of the nine slices, current chunk
    f32x4 xr0 = (f32x4){0.f, 0.f, 0.f, 0.f}, xr1 = xr0, xr2 = xr0;
#define VF_ULOAD(T, B, C)                                                                               \
    {                                                                                                   \
        const char* ub_ = (T).ubase + ((size_t)(C) * UCH + ((B) & ~1) * (FCO * FCK)) * 4;               \
        asm("" : "+s"(ub_), "+v"(uoffb));                                                               \
        ur##B = *(const __attribute__((address_space(1))) f32x4*)(                                      \
            (const __attribute__((address_space(1))) char*)ub_ + uoffb + ((B) & 1) * (FCO * FCK * 4));  \
    }
#define VF_ULOAD_ALL(T, C)                                                                              \
    { VF_ULOAD(T, 0, C); VF_ULOAD(T, 1, C); VF_ULOAD(T, 2, C); VF_ULOAD(T, 3, C); VF_ULOAD(T, 4, C);    \
      VF_ULOAD(T, 5, C); VF_ULOAD(T, 6, C); VF_ULOAD(T, 7, C); VF_ULOAD(T, 8, C); }
#define VF_XLOAD(T, C, R0, R1, R2) { fetch_x((T), 0, (C), R0); fetch_x((T), 1, (C), R1); fetch_x((T), 2, (C), R2); }
    // A window starts at pixel 4 t - 1, so pixel x sits at idx x + 5 (window = aligned b128 + b64) and a lane's four
    // pixels 4 q .. 4 q + 3 straddle two 16-byte slots.  Stored as four ds_write_b32 they hit 8 of the 32 store banks
    // (stride 4 dwords, channel planes 720 = 16 mod 32 apart): 4-way conflicts, 2/3 of this kernel's conflict cycles
    // (SQ_LDS_BANK_CONFLICT, round 4).  Instead every lane takes its LEFT neighbour's last pixel (DPP row_shr:1 -- the
    // first lane of a 16-lane row gets 0 = the left halo pixel; on 32-wide maps a DPP row holds two image rows and
    // lane 8 is zeroed by a select) and stores ONE aligned float4 (pixels 4 q - 1 .. 4 q + 2 at idx 4 q + 4: eight
    // consecutive lanes = 32 consecutive banks); the row's last pixel is one more ds_write_b32 from the row's last
    // lane only (EXEC set inside the asm statement: no branch for the compiler to lose its request count over).
#define VF_XST1(BUF, I, R)                                                                              \
    {                                                                                                   \
        int so_ = (BUF) * PSZ + VF_SCG(I) * CPS * G::PS + VF_SROW(I) * G::PW;                           \
        asm("" : "+s"(so_));                                                                            \
        float* d_ = Pl + so_ + xlofs;                                                                   \
        float pw_ = dpp_row_shr1(R.w);                                                                  \
        if (G::Q == 8) pw_ = xq == 0 ? 0.f : pw_;                                                       \
        *reinterpret_cast<f32x4*>(d_ + 4) = (f32x4){pw_, R.x, R.y, R.z};                                \
        /* (s_nop 4: an EXEC write needs five wait states before a DPP instruction -- the next slot's --, and the   \
           compiler cannot see this one) */                                                             \
        asm volatile("s_mov_b64 exec, %2\n\tds_write_b32 %0, %1 offset:32\n\ts_mov_b64 exec, -1\n\ts_nop 4" \
                     :: "v"((unsigned)(size_t)(__attribute__((address_space(3))) float*)d_), "v"(R.w), "s"(XLAST) : "memory"); \
    }
#define VF_XSTORE(T, BUF, R0, R1, R2) { VF_XST1(BUF, 0, R0); VF_XST1(BUF, 1, R1); VF_XST1(BUF, 2, R2); }

    // ---- input transform duty: channel tci, tile ttl, half th (wave-uniform): half 0 -> transformed rows 0-2 from
    // window rows 0-4, half 1 -> rows 3-5 from window rows 1-5.  The lanes of a ds_read_b128 lane group
    // ({0-3,12-15,20-27}, {4-11,16-19,28-31}, the same + 32) take 16 CONSECUTIVE tiles, whose 16-byte reads tile the 64 banks.
    const int th = wid >> 2;
    const int l5 = tid & 31;
    const bool gB = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    const int gpos = gB ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
    const int tci = (tid & 255) >> 5, ttl = (gB ? 16 : 0) + gpos;
    const int wpo = tci * G::PS + (4 * G::t_row(ttl) + th) * G::PW + 4 * G::t_col(ttl) + 4;
    // V[ci & 3][p 36][ci >> 2][tile 32]: the MFMA lane (tile li, k half lh) reads channel 4 lh + e of slice p at dword
    // (e * 36 + p) * 64 + lane
    const int vwo = ((tci & 3) * FNS + 18 * th) * 64 + (tci >> 2) * FTT + ttl;

    // V = B4^T d B4,  B4^T = [[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]]
    // window rows d0..d5 as column pairs (p0,p1),(p2,p3),(p4,p5).  Vertical pass, rows 0-2 (half 0):
    //   P = d4 - 4 d2;  t0 = 4 d0 + (P - d2);  R = d3 - 4 d1;  t1 = P + R;  t2 = P - R
    // rows 3-5 (half 1):  s = d3 - d1;  q = d4 - d2;  t3 = q + 2 s;  t4 = q - 2 s;  t5 = (d5 - d3) - 4 s
    f32x2 wr[5][3];                                      // the five window rows of this half: half 0 d0..d4, half 1 d1..d5
    f32x2 tv[3][3];                                      // the three transformed rows of this half
    // a window row = one ds_read_b128 (column pairs 0, 1) + one ds_read_b64 (pair 2); the pairs go through the
    // vertical pass separately, so the b64 halves are read when the b128 halves are already consumed
#define VF_WIN_READ4(BUF, J)                                                                            \
    {                                                                                                   \
        const f32x4 q_ = *reinterpret_cast<const f32x4*>(Pl + (BUF) * PSZ + wpo + (J) * G::PW);         \
        wr[J][0] = q_.xy; wr[J][1] = q_.zw;                                                             \
    }
#define VF_WIN_READ2(BUF, J) { wr[J][2] = *reinterpret_cast<const f32x2*>(Pl + (BUF) * PSZ + wpo + (J) * G::PW + 4); }
    auto win_rows1 = [&](int c, int th_) {               // th_: the half, a compile-time constant at every call site
        if (th_ == 0) {                                  // wr[j] = d_j
            const f32x2 P = pk_nmul4_add(wr[2][c], wr[4][c]);
            const f32x2 R = pk_nmul4_add(wr[1][c], wr[3][c]);
            tv[0][c] = pk_fmak<4>(wr[0][c], pk_sub(P, wr[2][c]));
            tv[1][c] = pk_add(P, R);
            tv[2][c] = pk_sub(P, R);
        } else {                                         // wr[j] = d_{j+1}
            const f32x2 s = pk_sub(wr[2][c], wr[0][c]);
            const f32x2 q = pk_sub(wr[3][c], wr[1][c]);
            tv[0][c] = pk_fmak<2>(s, q);
            f32x2 t4;
            asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=v"(t4) : "v"(s), "v"(q));
            tv[1][c] = t4;                               // q - 2 s
            tv[2][c] = pk_nmul4_add(s, pk_sub(wr[4][c], wr[2][c]));
        }
    };
    // columns of one transformed row (p0,p1),(p2,p3),(p4,p5) -> c0..c5:
    //   P2 = (p4,p5) - 4 (p2,p3);  [c0, c5] = 4 (p0,p1) + (P2 - (p2,p3));  Q = (p2,p3) - 4 (p0,p1);  [c1, c2] = P2.x +- Q.y
    //   E = (p4,p5) - (p2,p3);  F = (p2,p3) - (p0,p1);  [c3, c4] = E.x +- 2 F.y
    f32x2 c05, c12, c34;
    auto win_col = [&](int r) {
        const f32x2 p01 = tv[r][0], p23 = tv[r][1], p45 = tv[r][2];
        const f32x2 P2 = pk_nmul4_add(p23, p45);
        c05 = pk_fmak<4>(p01, pk_sub(P2, p23));
        const f32x2 Q = pk_nmul4_add(p01, p23);
        c12 = pk_lo_pm_hi(P2, Q);
        const f32x2 E = pk_sub(p45, p23), F = pk_sub(p23, p01);
        asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_hi:[1,0,0]" : "=v"(c34) : "v"(F), "v"(E));
    };
    // transformed row r of this half (a = 3 th + r): p offsets relative to the half's first slot
#define VF_WIN_WRITE(BUF, R_)                                                                           \
    {                                                                                                   \
        float* vo_ = Vl + (BUF) * VSZ + vwo;                                                            \
        vo_[f44_p(R_, 0) * 64] = c05.x; vo_[f44_p(R_, 1) * 64] = c12.x; vo_[f44_p(R_, 2) * 64] = c12.y;  \
        vo_[f44_p(R_, 3) * 64] = c34.x; vo_[f44_p(R_, 4) * 64] = c34.y; vo_[f44_p(R_, 5) * 64] = c05.y;  \
    }
    // (one base per V buffer: every fragment is then base + a multiple of 64 dwords below 256 -- ds_read2st64_b32)
    const float* const vb0 = Vl + (wg_ * 9) * 64 + lane;
    const float* const vb1 = vb0 + VSZ;                  // (the compiler folds this back into offsets: an opaque second
                                                         // base register was tried and tips the kernel into spills)

    // ---- first loads of the first tile: U(0), rows(0), rows(1) -- all issued together (one round trip)
    f32x4 yr0 = xr0, yr1 = xr0, yr2 = xr0;
    VF_ULOAD_ALL(cur, 0);
    VF_XLOAD(cur, 0, xr0, xr1, xr2);
    VF_XLOAD(cur, min(1, clast), yr0, yr1, yr2);
    for (int i = tid; i < 2 * PSZ; i += NT_) Pl[i] = 0.f;    // halo columns stay zero in both buffers, for every tile

#ifdef VF_STAMPS44F
    const unsigned long long rt_0 = __builtin_amdgcn_s_memrealtime(), ct_0 = __builtin_amdgcn_s_memtime();
    unsigned long long st_loop = 0, st_epi = 0, st_tiles = 0, st_e[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_p[5];
#define VF_STAMP(V) const unsigned long long V = __builtin_amdgcn_s_memtime()
#else
#define VF_STAMP(V)
#endif
    // One slice = four MFMAs (K = 8) on accumulator B; side work goes BEHIND the slice's own MFMAs.  FIRST: the
    // accumulator starts from the literal 0.
#ifdef VF_AB_NOU
#define VF_ABU(...)
#else
#define VF_ABU(...) __VA_ARGS__
#endif
#define VF_SLICE(C, FIRST, B, BC, SIDE0, SIDE1, SIDE2)                                                   \
    {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc[B] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur##B.x, BC[0], (FIRST) ? (f32x16){0} : acc[B], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        SIDE0;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc[B] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur##B.y, BC[1], acc[B], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        SIDE1;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc[B] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur##B.z, BC[2], acc[B], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        SIDE2;                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        acc[B] = __builtin_amdgcn_mfma_f32_32x32x2f32(ur##B.w, BC[3], acc[B], 0, 0, 0);                  \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        VF_ABU(VF_ULOAD(cur, B, min((C) + 1, clast)));    /* this slice's U of the NEXT chunk, in place */ \
    }
#define VF_BFRAG(D, BUF_, S_)                                                                            \
    { _Pragma("unroll") for (int e = 0; e < 4; ++e) D[e] = ((BUF_) ? vb1 : vb0)[((S_) + e * FNS) * 64]; }
    // One chunk, PAR = chunk parity = its V buffer (compile time).  The B fragments alternate between two register
    // sets X, Y (nine slices: the NEXT chunk starts on the other set, which is why the sets are macro arguments);
    // slice 0's are loaded by the previous chunk.
    //   slice 0: b128 halves of the window rows of chunk C+1 from P[PAR^1]     slice 1: vertical pass on them, b64 halves
    //   slices 2-4: rest of the vertical pass, horizontal pass + 6 values -> V[PAR^1] per transformed row
    //   slice 6: raw rows of chunk C+2 -> P[PAR]
    //   ... the chunk's ONE barrier sits behind slice 7, with the fragments of slice 8 already in registers:
    //   slice 8: raw rows of chunk C+3 requested, slice-0 fragments of chunk C+1 from V[PAR^1]
    // Hazards: V[PAR] is rewritten in slices 2-4 of chunk C+1 (behind this barrier, all its reads are in front);
    // P[PAR^1] is rewritten in slice 6 of chunk C+1, read in slices 0-1 of C.  Staging indices beyond the last chunk
    // are clamped (the final iterations redo harmless loads / LDS writes that nobody reads).
#ifdef VF_AB_NOXFORM      /* diagnostic builds (tools/wino44f_ablate.sh): leave one kind of side work out of the chunk loop */
#define VF_ABX(...)
#else
#define VF_ABX(...) __VA_ARGS__
#endif
#ifdef VF_AB_NOX
#define VF_ABR(...)
#else
#define VF_ABR(...) __VA_ARGS__
#endif
#ifdef VF_AB_NOBF
#define VF_ABB(...)
#else
#define VF_ABB(...) __VA_ARGS__
#endif
#define VF_CHUNK(C, PAR, FIRST, X, Y, TH)                                                                \
    {                                                                                                    \
        VF_SLICE(C, FIRST, 0, X, { VF_ABB(VF_BFRAG(Y, PAR, 1)); VF_ABX(VF_WIN_READ4((PAR) ^ 1, 0); VF_WIN_READ4((PAR) ^ 1, 1)); }, \
                 { VF_ABX(VF_WIN_READ4((PAR) ^ 1, 2); VF_WIN_READ4((PAR) ^ 1, 3)); }, VF_ABX(VF_WIN_READ4((PAR) ^ 1, 4))); \
        VF_SLICE(C, FIRST, 1, Y, { VF_ABB(VF_BFRAG(X, PAR, 2)); VF_ABX(win_rows1(0, TH)); }, VF_ABX(win_rows1(1, TH)), \
                 { VF_ABX(VF_WIN_READ2((PAR) ^ 1, 0); VF_WIN_READ2((PAR) ^ 1, 1); VF_WIN_READ2((PAR) ^ 1, 2); \
                   VF_WIN_READ2((PAR) ^ 1, 3); VF_WIN_READ2((PAR) ^ 1, 4)); });                          \
        VF_SLICE(C, FIRST, 2, X, { VF_ABB(VF_BFRAG(Y, PAR, 3)); VF_ABX(win_rows1(2, TH)); }, VF_ABX(win_col(0)), VF_ABX(VF_WIN_WRITE((PAR) ^ 1, 0))); \
        VF_SLICE(C, FIRST, 3, Y, { VF_ABB(VF_BFRAG(X, PAR, 4)); VF_ABX(win_col(1)); }, VF_ABX(VF_WIN_WRITE((PAR) ^ 1, 1)), (void)0); \
        VF_SLICE(C, FIRST, 4, X, { VF_ABB(VF_BFRAG(Y, PAR, 5)); VF_ABX(win_col(2)); }, VF_ABX(VF_WIN_WRITE((PAR) ^ 1, 2)), (void)0); \
        VF_SLICE(C, FIRST, 5, Y, VF_ABB(VF_BFRAG(X, PAR, 6)), (void)0, (void)0);                         \
        VF_SLICE(C, FIRST, 6, X, VF_ABB(VF_BFRAG(Y, PAR, 7)), VF_ABR(VF_XSTORE(cur, PAR, xr0, xr1, xr2)), (void)0); \
        VF_SLICE(C, FIRST, 7, Y, VF_ABB(VF_BFRAG(X, PAR, 8)), (void)0, (void)0);                         \
        VF_LDS_BARRIER();                                                                                \
        VF_SLICE(C, FIRST, 8, X, VF_ABB(VF_BFRAG(Y, (PAR) ^ 1, 0)), VF_ABR(VF_XLOAD(cur, min((C) + 3, clast), xr0, xr1, xr2)), (void)0); \
    }

    // Staging of a tile's first chunks: rows(0), rows(1) -> P[0], P[1] (rows(2) requested), then V(0) from P[0].
#define VF_STAGE_ROWS(T)                                                                                 \
    {                                                                                                    \
        VF_XSTORE(T, 0, xr0, xr1, xr2);                                                                  \
        VF_XSTORE(T, 1, yr0, yr1, yr2);                                                                  \
        VF_XLOAD(T, min(2, clast), xr0, xr1, xr2);                                                       \
    }
#define VF_STAGE_V0()                                                                                    \
    {                                                                                                    \
        VF_WIN_READ4(0, 0); VF_WIN_READ4(0, 1); VF_WIN_READ4(0, 2); VF_WIN_READ4(0, 3); VF_WIN_READ4(0, 4); \
        VF_WIN_READ2(0, 0); VF_WIN_READ2(0, 1); VF_WIN_READ2(0, 2); VF_WIN_READ2(0, 3); VF_WIN_READ2(0, 4); \
        if (th == 0) { win_rows1(0, 0); win_rows1(1, 0); win_rows1(2, 0); }                              \
        else { win_rows1(0, 1); win_rows1(1, 1); win_rows1(2, 1); }                                      \
        win_col(0); VF_WIN_WRITE(0, 0); win_col(1); VF_WIN_WRITE(0, 1); win_col(2); VF_WIN_WRITE(0, 2);  \
    }
    __syncthreads();                                      // zero fill done
    VF_STAGE_ROWS(cur);
    __syncthreads();
    VF_STAGE_V0();
    __syncthreads();

    for (;;) {
        f32x16 acc[9];
        float bfA[4], bfB[4];
        VF_BFRAG(bfA, 0, 0);
        VF_STAMP(t_1);

        // (the two halves of the transform duty run their own copy of the loop: the half is a compile-time constant there)
#define VF_LOOP(TH)                                                                                      \
        {                                                                                                \
            VF_CHUNK(0, 0, 1, bfA, bfB, TH);                                                             \
            int c = 1;                                                                                   \
            for (; c + 1 < nch; c += 2) {                                                                \
                VF_CHUNK(c, 1, 0, bfB, bfA, TH);                                                         \
                VF_CHUNK(c + 1, 0, 0, bfA, bfB, TH);                                                     \
            }                                                                                            \
            if (c < nch) VF_CHUNK(c, 1, 0, bfB, bfA, TH);                                                \
        }
        if (nch <= 0) {
#pragma unroll
            for (int b = 0; b < 9; ++b) acc[b] = (f32x16){0};
        } else if (th == 0) {
            VF_LOOP(0);
        } else {
            VF_LOOP(1);
        }
#undef VF_LOOP

        VF_STAMP(t_2);
        const unsigned lin_next = lin + F44_PERSIST;
        const bool has_next = !partial && lin_next < (unsigned)a.nfull;      // workgroup-uniform
        const unsigned logical_cur = logical_of(lin);
        int s, r0, cot_;
        tile_pos(logical_cur, s, r0, cot_);
        const int co0 = cot_ * FCO;

        // The epilogue's per-lane index arithmetic is the same for every tile; an opaque copy of the lane id keeps the
        // compiler from hoisting all of it out of the tile loop (and spilling it across the chunk loop).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, lh_e = lane_e >> 5;
        const int orow = r0 + 4 * G::t_row(li_e), ocol = 4 * G::t_col(li_e);
        float* const xch = Vl;
        // the next whole tile's first loads go out FIRST (U fragments and raw-row registers are dead; issued
        // unconditionally -- the last tile re-reads its own first chunks): every output store of this epilogue is then
        // younger than they are, and no wait for them (in-order counter) has to sit out a store
        const Tile nx = make_tile(has_next ? logical_of(lin_next) : logical_cur);
        VF_ULOAD_ALL(nx, 0);
        VF_XLOAD(nx, 0, xr0, xr1, xr2);
        VF_XLOAD(nx, min(1, clast), yr0, yr1, yr2);

        // ---- output transform.  Columns (A4^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]) in
        // registers: the full row M[0..5] = acc[0..5] gives T[0..3]; the half row acc[6..8] gives the three values its
        // partner needs: cols 0-2: (m0 + m1 + m2, m1 - m2, m1 + m2), cols 3-5: (m3 + m4, m3 - m4, m5).  Rows (the same
        // A4^T) after the exchange through LDS.  FOUR passes of four accumulator registers through TWO exchange buffers:
        // while the waves publish pass k + 1 they finish pass k (LDS reads, row transform, operands, output stores), so
        // LDS writes, LDS reads, arithmetic and stores of different waves overlap and every phase has ONE barrier:
        //   publish 0 | publish 1, finish 0 | publish 2, finish 1 | publish 3, finish 2, stage the next tile's rows |
        //   finish 3, V(0) of the next tile (its slot overlays exchange buffer 0 only, free since the previous barrier)
        // Wave (cw, g) finishes register 4 k + g of its channel half in pass k: channel co0 + 32 cw + 8 k + 4 lh + g.
        const bool jh = wg_ & 1;                             // (uniform) which half of the shared row
        constexpr int XB = 8 * 7 * 4 * 64;                   // floats of one exchange buffer: [wave 8][value 7][reg 4][lane 64]
        static_assert(2 * XB <= XCH && VSZ <= XB, "exchange buffers");
        // epilogue operands of this lane's four channels: requested now, used from the second phase on
        const int cob = co0 + cw * 32 + 4 * lh_e + wg_;      // + 8 k
        float ebias[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ebias[k] = 0.f;
        if (!partial) {
            if (a.bias) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ebias[k] = a.bias[min(cob + 8 * k, a.Cout - 1)];
            }
            if (a.vbias) {
#pragma unroll
                for (int k = 0; k < 4; ++k) ebias[k] += a.vbias[(size_t)s * a.Cout + min(cob + 8 * k, a.Cout - 1)];
            }
        }
        float4 er[4];                                        // residual rows of the pass that is finished NEXT phase
#pragma unroll
        for (int y = 0; y < 4; ++y) er[y] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ph = 0; ph < 5; ++ph) {
            if (ph < 4) {                                    // ---- publish pass ph into buffer ph & 1
                float* xo = xch + (ph & 1) * XB + (size_t)(wid * 7) * (4 * 64) + lane_e;
#pragma unroll
                for (int k = 0; k < 2; ++k) {                // register pairs (4 ph + 2 k, + 1)
                    const int r = 4 * ph + 2 * k;
                    const f32x2 m0 = (f32x2){acc[0][r], acc[0][r + 1]}, m1 = (f32x2){acc[1][r], acc[1][r + 1]};
                    const f32x2 m2 = (f32x2){acc[2][r], acc[2][r + 1]}, m3 = (f32x2){acc[3][r], acc[3][r + 1]};
                    const f32x2 m4 = (f32x2){acc[4][r], acc[4][r + 1]}, m5 = (f32x2){acc[5][r], acc[5][r + 1]};
                    const f32x2 s12 = pk_add(m1, m2), d12 = pk_sub(m1, m2), s34 = pk_add(m3, m4), d34 = pk_sub(m3, m4);
                    const f32x2 T0 = pk_add(pk_add(m0, s12), s34);
                    const f32x2 T1 = pk_fmak<2>(d34, d12);
                    const f32x2 T2 = pk_fmak<4>(s34, s12);
                    const f32x2 T3 = pk_add(pk_fmak<8>(d34, d12), m5);
                    const f32x2 h0 = (f32x2){acc[6][r], acc[6][r + 1]}, h1 = (f32x2){acc[7][r], acc[7][r + 1]};
                    const f32x2 h2 = (f32x2){acc[8][r], acc[8][r + 1]};
                    f32x2 H0, H1, H2;
                    if (jh) {                                // cols 3-5: (m3 + m4, m3 - m4, m5)
                        H0 = pk_add(h0, h1); H1 = pk_sub(h0, h1); H2 = h2;
                    } else {                                 // cols 0-2: (m0 + (m1 + m2), m1 - m2, m1 + m2)
                        H2 = pk_add(h1, h2); H1 = pk_sub(h1, h2); H0 = pk_add(h0, H2);
                    }
                    xo[(0 * 4 + 2 * k) * 64] = T0.x; xo[(0 * 4 + 2 * k + 1) * 64] = T0.y;
                    xo[(1 * 4 + 2 * k) * 64] = T1.x; xo[(1 * 4 + 2 * k + 1) * 64] = T1.y;
                    xo[(2 * 4 + 2 * k) * 64] = T2.x; xo[(2 * 4 + 2 * k + 1) * 64] = T2.y;
                    xo[(3 * 4 + 2 * k) * 64] = T3.x; xo[(3 * 4 + 2 * k + 1) * 64] = T3.y;
                    xo[(4 * 4 + 2 * k) * 64] = H0.x; xo[(4 * 4 + 2 * k + 1) * 64] = H0.y;
                    xo[(5 * 4 + 2 * k) * 64] = H1.x; xo[(5 * 4 + 2 * k + 1) * 64] = H1.y;
                    xo[(6 * 4 + 2 * k) * 64] = H2.x; xo[(6 * 4 + 2 * k + 1) * 64] = H2.y;
                }
            }
            if (ph > 0) {                                    // ---- finish pass ph - 1 from buffer (ph - 1) & 1
                constexpr int GS = 2 * 7 * 4 * 64, VS = 4 * 64;      // stride between slice groups (waves 2 g' + cw) / values
                const int co = cob + 8 * (ph - 1);
                const float* xi = xch + ((ph - 1) & 1) * XB + (size_t)(cw * 7) * (4 * 64) + wg_ * 64 + lane_e;
                float T[6][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    T[0][c] = xi[0 * GS + c * VS];
                    T[2][c] = xi[1 * GS + c * VS];
                    T[3][c] = xi[2 * GS + c * VS];
                    T[5][c] = xi[3 * GS + c * VS];
                }
                {
                    const float A_ = xi[0 * GS + 4 * VS], B_ = xi[0 * GS + 5 * VS], C_ = xi[0 * GS + 6 * VS];
                    const float D_ = xi[1 * GS + 4 * VS], E_ = xi[1 * GS + 5 * VS], F_ = xi[1 * GS + 6 * VS];
                    T[1][0] = A_ + D_; T[1][1] = __builtin_fmaf(2.f, E_, B_); T[1][2] = __builtin_fmaf(4.f, D_, C_);
                    T[1][3] = __builtin_fmaf(8.f, E_, B_) + F_;
                }
                {
                    const float A_ = xi[2 * GS + 4 * VS], B_ = xi[2 * GS + 5 * VS], C_ = xi[2 * GS + 6 * VS];
                    const float D_ = xi[3 * GS + 4 * VS], E_ = xi[3 * GS + 5 * VS], F_ = xi[3 * GS + 6 * VS];
                    T[4][0] = A_ + D_; T[4][1] = __builtin_fmaf(2.f, E_, B_); T[4][2] = __builtin_fmaf(4.f, D_, C_);
                    T[4][3] = __builtin_fmaf(8.f, E_, B_) + F_;
                }
                float Y[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float S12 = T[1][c] + T[2][c], D12 = T[1][c] - T[2][c];
                    const float S34 = T[3][c] + T[4][c], D34 = T[3][c] - T[4][c];
                    Y[0][c] = (T[0][c] + S12) + S34;
                    Y[1][c] = __builtin_fmaf(2.f, D34, D12);
                    Y[2][c] = __builtin_fmaf(4.f, S34, S12);
                    Y[3][c] = __builtin_fmaf(8.f, D34, D12) + T[5][c];
                }
                if (partial) {                               // raw partial tile: ws[part][co 64][tile 32][4x4]
                    float* w16 = a.ws + (((size_t)tail_id * FCO + (co - co0)) * FTT + li_e) * 16;
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        *reinterpret_cast<float4*>(w16 + 4 * y) = make_float4(Y[y][0], Y[y][1], Y[y][2], Y[y][3]);
                } else if (co < a.Cout) {
                    const float bb = ebias[ph - 1];
                    const size_t o = ((size_t)s * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
#pragma unroll
                    for (int y = 0; y < 4; ++y)
                        *reinterpret_cast<float4*>(a.y + o + y * G::W) =
                            make_float4(Y[y][0] + bb + er[y].x, Y[y][1] + bb + er[y].y, Y[y][2] + bb + er[y].z,
                                        Y[y][3] + bb + er[y].w);
                }
            }
            if (ph < 4 && !partial && a.res) {               // residual rows of pass ph: used in the next phase
                const int co = min(cob + 8 * ph, a.Cout - 1);
                const size_t o = ((size_t)s * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
#pragma unroll
                for (int y = 0; y < 4; ++y) er[y] = *reinterpret_cast<const float4*>(a.res + o + y * G::W);
            }
            if (ph == 3 && has_next) VF_STAGE_ROWS(nx);      // (P is outside the exchange area)
            if (ph == 4 && has_next) VF_STAGE_V0();
#ifdef VF_STAMPS44F
            t_p[ph] = __builtin_amdgcn_s_memtime();
#endif
            if (ph < 4 || has_next) VF_LDS_BARRIER();
        }
        VF_STAMP(t_3);
#ifdef VF_STAMPS44F
        st_e[2] += t_p[0] - t_2; st_e[3] += t_p[1] - t_p[0]; st_e[4] += t_p[2] - t_p[1];
        st_e[5] += t_p[3] - t_p[2]; st_e[6] += t_p[4] - t_p[3];
        st_loop += t_2 - t_1; st_epi += t_3 - t_2; st_tiles += 1; st_e[0] += t_3 - t_2;
#endif
        if (!has_next) break;
        cur = nx;
        lin = lin_next;
    }
#ifdef VF_STAMPS44F
    if (tid == 0 && !partial) {
        atomicAdd(&g_stamps44f[0], st_loop); atomicAdd(&g_stamps44f[1], st_epi); atomicAdd(&g_stamps44f[2], st_tiles);
        for (int i = 0; i < 7; ++i) atomicAdd(&g_stamps44f[3 + i], st_e[i]);
        atomicAdd(&g_stamps44f[10], __builtin_amdgcn_s_memtime() - ct_0);            // shader clocks ...
        atomicAdd(&g_stamps44f[11], __builtin_amdgcn_s_memrealtime() - rt_0);        // ... per 100 MHz ticks
    }
#endif
#undef VF_STAMP
#undef VF_ULOAD
#undef VF_ULOAD_ALL
#undef VF_XLOAD
#undef VF_XST1
#undef VF_XSTORE
#undef VF_SID
#undef VF_SROW
#undef VF_SCG
#undef VF_SUY
#undef VF_STAGE_ROWS
#undef VF_STAGE_V0
#undef VF_SLICE
#undef VF_BFRAG
#undef VF_CHUNK
#undef VF_WIN_READ4
#undef VF_WIN_READ2
#undef VF_WIN_WRITE
}

// OIHW -> transformed + packed forward  U[co tile][ci chunk][p 36][co 64][ci 8] = (G4 w G4^T) at slice (a, b), p = f44_p(a, b),
//        and backward (dgrad)          [ci tile][co chunk][p][ci 64][co 8] of the 180-degree-rotated kernel.
// One 512-thread workgroup per (tile, chunk) group = 18432 outputs: thread (m, k8) reads the nine taps of one
// (co, ci) pair once and writes its 36 slices, each slice a contiguous 2 KB line of the workgroup.  Evaluated in double.
__device__ __forceinline__ void f44_pack_group(const float* __restrict__ w, float* __restrict__ uf,
                                               float* __restrict__ ub, int Cout, int Cin, size_t nf, size_t nb,
                                               size_t group) {
    constexpr int GSZ = FNS * FCO * FCK;
    const size_t ngf = nf / GSZ;
    const bool bwd = group >= ngf;
    if (bwd) {
        group -= ngf;
        if (group >= nb / GSZ || !ub) return;
    }
    const int M = bwd ? Cin : Cout, K = bwd ? Cout : Cin;
    const int nchunk = (K + FCK - 1) / FCK;
    const int chunk = group % nchunk, mt = group / nchunk;
    const int t = threadIdx.x;
    const int m = t >> 3, k8 = t & 7;
    const int mm = mt * 64 + m, kk = chunk * FCK + k8;
    double g[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = 0.0;
    if (mm < M && kk < K) {
        const int co = bwd ? kk : mm, ci = bwd ? mm : kk;
        const float* p = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = (double)(bwd ? p[8 - i] : p[i]);
    }
    // G4 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    auto g4 = [](double x0, double x1, double x2, double* o) {
        const double a6 = -(x0 + x2) / 6.0, b6 = x1 / 6.0;
        const double a24 = x0 / 24.0 + x2 / 6.0, b12 = x1 / 12.0;
        o[0] = x0 / 4.0; o[1] = a6 - b6; o[2] = a6 + b6; o[3] = a24 + b12; o[4] = a24 - b12; o[5] = x2;
    };
    double tq[6][3];                                  // rows: G4 applied to the kernel rows
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        double o[6];
        g4(g[q], g[3 + q], g[6 + q], o);
#pragma unroll
        for (int i = 0; i < 6; ++i) tq[i][q] = o[i];
    }
    float* out = (bwd ? ub : uf) + group * (size_t)GSZ + t;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        double o[6];
        g4(tq[i][0], tq[i][1], tq[i][2], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) out[f44_p(i, j) * (FCO * FCK)] = (float)o[j];
    }
}

__global__ __launch_bounds__(512) void wino44f_pack_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                           float* __restrict__ ub, int Cout, int Cin, size_t nf,
                                                           size_t nb) {
    f44_pack_group(w, uf, ub, Cout, Cin, nf, nb, blockIdx.x);
}

constexpr int F44_PACK_BLOCKS = FNS * FCO * FCK / 256;     // 256-output units per pack group (72)

struct F44PackDesc {
    const float* w;
    float* uf;
    float* ub;
    long long Cout, Cin, nf, nb, first_block;         // first_block in units of 256 outputs (72 per group)
};
__global__ __launch_bounds__(512) void wino44f_pack_multi_kernel(const F44PackDesc* __restrict__ desc, int nlayers) {
    const long long vb = (long long)blockIdx.x * F44_PACK_BLOCKS;
    int lo = 0, hi = nlayers;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= vb) lo = mid; else hi = mid;
    }
    const F44PackDesc d = desc[lo];
    f44_pack_group(d.w, d.uf, d.ub, (int)d.Cout, (int)d.Cin, (size_t)d.nf, (size_t)d.nb,
                   (size_t)((vb - d.first_block) / F44_PACK_BLOCKS));
}

inline int rupf(int v, int m) { return (v + m - 1) / m * m; }

// Sums the K-range partials of the tail tiles in a fixed order and applies the epilogue (bias + per-view bias +
// residual).  One thread per (tail tile, co, 4x4 tile, tile row).
template <int LOGW>
__global__ __launch_bounds__(256) void wino44_fixup_kernel(F44Args a, int ntail) {
    using G = FGeo<LOGW, 0>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int yy = idx & 3;
    const int tl = (idx >> 2) % FTT;
    const int col = (idx / (4 * FTT)) % FCO;
    const int j = idx / (4 * FTT * FCO);
    if (j >= ntail) return;
    const int ncot = a.CoutP / FCO;
    const int logical = a.nfull + j;
    const int cot = logical % ncot, wg = logical / ncot;
    const int s = wg / G::WPI, r0 = (wg % G::WPI) * 4 * G::TR;
    const int co = cot * FCO + col;
    if (co >= a.Cout) return;
    const int orow = r0 + 4 * G::t_row(tl) + yy, ocol = 4 * G::t_col(tl);
    const size_t o = ((size_t)s * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
    float4 t[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
        t[p] = *reinterpret_cast<const float4*>(
            a.ws + ((((size_t)j * a.tail_split + min(p, a.tail_split - 1)) * FCO + col) * FTT + tl) * 16 + 4 * yy);
    float b = 0.f;
    if (a.bias) b += a.bias[co];
    if (a.vbias) b += a.vbias[(size_t)s * a.Cout + co];
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.res) q = *reinterpret_cast<const float4*>(a.res + o);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < a.tail_split) { v.x += t[p].x; v.y += t[p].y; v.z += t[p].z; v.w += t[p].w; }
    }
    v.x += b; v.y += b; v.z += b; v.w += b;
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
    *reinterpret_cast<float4*>(a.y + o) = v;
}

inline int f44_groups(int S, int H, int W) { return S * ((H / 4) * (W / 4) / FTT); }

template <int LOGW, int MODE, bool RAGGED>
int launch_f44r(F44Args a, size_t ws_floats, hipStream_t st) {
    using G = FGeo<LOGW, MODE>;
    const int T = G::groups(a.S) * (a.CoutP / FCO);
    wino_tail_plan(T, a.CinP / FCK, &a.nfull, &a.tail_split);
    const int ntail = T - a.nfull;
    if ((size_t)ntail * a.tail_split * FCO * FTT * 16 > ws_floats || !a.ws) {   // no room: plain grid
        a.nfull = T;
        a.tail_split = 1;
    }
    const int nt = T - a.nfull;
    a.npers = a.nfull < F44_PERSIST ? a.nfull : F44_PERSIST;
    hipLaunchKernelGGL((wino44_conv_kernel<LOGW, MODE, RAGGED>), dim3(a.npers + nt * a.tail_split), dim3(512), 0, st, a);
    if (nt > 0)
        hipLaunchKernelGGL((wino44_fixup_kernel<LOGW>), dim3((nt * FCO * FTT * 4 + 255) / 256), dim3(256), 0, st, a, nt);
    VF_RETURN_LAST_ERROR();
}

template <int LOGW, int MODE>
int launch_f44(F44Args a, size_t ws_floats, hipStream_t st) {
    return (a.Cin & (FCK - 1)) ? launch_f44r<LOGW, MODE, true>(a, ws_floats, st) : launch_f44r<LOGW, MODE, false>(a, ws_floats, st);
}

}  // namespace

extern "C" {

int vf_wino44_pack_sizes(int Cout, int Cin, long* fwd_floats, long* bwd_floats) {
    *fwd_floats = (long)FNS * rupf(Cin, FCK) * rupf(Cout, FCO);
    *bwd_floats = (long)FNS * rupf(Cout, FCK) * rupf(Cin, FCO);
    return 0;
}

int vf_wino44_pack_weights(const float* w_oihw, float* u_fwd, float* u_bwd, int Cout, int Cin, void* stream) {
    const size_t nf = (size_t)FNS * rupf(Cin, FCK) * rupf(Cout, FCO);
    const size_t nb = u_bwd ? (size_t)FNS * rupf(Cout, FCK) * rupf(Cin, FCO) : 0;
    hipLaunchKernelGGL(wino44f_pack_kernel, dim3((unsigned)((nf + nb) / (FNS * FCO * FCK))), dim3(512), 0,
                       (hipStream_t)stream, w_oihw, u_fwd, u_bwd, Cout, Cin, nf, nb);
    VF_RETURN_LAST_ERROR();
}

// desc: device int64 [nlayers][8] rows {w, u_fwd, u_bwd, Cout, Cin, fwd_floats, bwd_floats, first_block}
int vf_wino44_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream) {
    if (nlayers <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(wino44f_pack_multi_kernel, dim3((unsigned)(total_blocks / F44_PACK_BLOCKS)), dim3(512), 0,
                       (hipStream_t)stream, (const F44PackDesc*)desc, nlayers);
    VF_RETURN_LAST_ERROR();
}

// 1 if vf_wino44_conv_fwd supports this (output) size / mode: 3x3 stride 1, H = W in {32, 64}, modes 0 / 2.
int vf_wino44_supported(int H, int W, int mode) {
    return (H == W && (W == 32 || W == 64) && (mode == 0 || mode == 2)) ? 1 : 0;
}

// workspace floats vf_wino44_conv_fwd wants for its split tail tiles (0 when the grid divides evenly)
long vf_wino44_conv_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const int T = f44_groups(S, H, W) * (rupf(Cout, FCO) / FCO);
    int nfull, split;
    wino_tail_plan(T, rupf(Cin, FCK) / FCK, &nfull, &split);
    return (long)(T - nfull) * split * FCO * FTT * 16;
}

// Expected CU fill (percent) of vf_wino44_conv_fwd at this shape, and the tile count (as vf_wino_conv_fill_pct).
int vf_wino44_conv_fill_pct(int S, int Cin, int Cout, int H, int W, int* tiles_out) {
    const int T = f44_groups(S, H, W) * (rupf(Cout, FCO) / FCO);
    const int nch = rupf(Cin, FCK) / FCK;
    int nfull, split;
    wino_tail_plan(T, nch, &nfull, &split, 0.0);
    if (tiles_out) *tiles_out = T;
    if (T <= 0) return 0;
    const double time = (nfull + WINO_SLOTS - 1) / WINO_SLOTS + (T > nfull ? wino_tail_time(T - nfull, split, nch, 0.0) : 0.0);
    return (int)(100.0 * T / WINO_SLOTS / time);
}

// y = conv3x3(x) (+bias +view_bias +residual), pad 1, stride 1, via the fused Winograd F(4x4,3x3).
// u_packed from vf_wino44_pack_weights (forward pack for the conv, backward pack for its dgrad).
int vf_wino44_conv_fwd(const float* x, const float* u_packed, const float* bias, const float* view_bias,
                       const float* residual, float* y, float* ws, long ws_floats, int S, int Cin, int Cout, int H,
                       int W, int mode, void* stream) {
    if (S <= 0) return 0;
    if (!vf_wino44_supported(H, W, mode)) return (int)hipErrorInvalidValue;
    F44Args a;
    a.x = x; a.u = u_packed; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.S = S; a.Cin = Cin; a.Cout = Cout; a.CinP = rupf(Cin, FCK); a.CoutP = rupf(Cout, FCO);
    a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    const size_t nws = ws ? (size_t)ws_floats : 0;
    if (W == 32) return mode == 0 ? launch_f44<5, 0>(a, nws, st) : launch_f44<5, 2>(a, nws, st);
    return mode == 0 ? launch_f44<6, 0>(a, nws, st) : launch_f44<6, 2>(a, nws, st);
}

}  // extern "C"

#ifdef VF_STAMPS44F
extern "C" void vf_debug_stamps44f(unsigned long long* out16, int reset) {
    if (out16) (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps44f), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps44f), z, 128); }
}
#endif
