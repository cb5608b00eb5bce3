// Fused Winograd convolution, forward and dgrad, for the stride-1 3x3 layers (reference model/unet.py:42,189,214):
// the NESTED minimal filtering F(2,3) x F(4,3) -- two output rows by four output columns per tile:
//
//     Y (2x4) = A2^T [ sum_ci (G2 g G4^T) (.) (B2^T d B4) ] A4        d = 4x6 input window, 24 products per 8 outputs
//
// i.e. 3 multiplies per output pixel and (ci, co) pair instead of 4 for F(2x2,3x3) and 9 for the direct form.
// fp32 throughout.  Per-layer error against an fp64 convolution (random data, K = 576 ... 2880): rel-L2 0.7-1.4e-6
// (F(2x2): 0.3-0.6e-6, direct fp32: 0.2-0.3e-6, F(4x4): 1.7-3.7e-6) -- the 4-point transform is applied along ONE
// axis only, so its constants (4, 5, 8, 1/6, 1/24) enter once, not squared.  The weight transform is evaluated in
// double and rounded once.
//
// Everything is fused in one kernel -- no transformed tensor ever goes to HBM:
//   * weights arrive pre-transformed + packed  U[co tile][chunk][slice 24][co 64][ci 8]  (pack kernel).  A wave's
//     MFMA A operands are read STRAIGHT from that image into registers (one float4 per lane and slice: the lane's
//     four k values): with all 32 tiles of the workgroup in one wave no two waves share a U element, so staging U
//     through LDS (a third of the F(2x2) kernel's LDS traffic and 64 KB of its LDS) would only re-shuffle it;
//   * per 8-channel chunk the raw haloed rows are staged in LDS, two threads per (channel, tile) transform the 4x6
//     window (B2^T d B4) into the V[slice][ci][tile] image -- three 16-byte reads per window row, conflict-free;
//   * 24 independent GEMM slices  D_s[co][tile] += U_s[co][ci] V_s[ci][tile]  on v_mfma_f32_32x32x2_f32.  Workgroup =
//     8 waves = 64 co x 32 tiles (256 output pixels); wave (cw, a) owns 32 co x 32 tiles x the 6 slices of transformed
//     row a = 96 accumulators.  ONE workgroup per CU, persistent over the whole tiles (next tile's first loads issued
//     under the epilogue);
//   * the epilogue applies A4 per wave in registers, the four row waves exchange their 2x4 partial tiles through LDS
//     once (A2), every wave finishes a quarter of the channels: bias + per-view bias + residual, float4 row stores.
// MODE 0: plain input; MODE 2: nearest-x2-upsampled input (Upsample conv), as in conv.hip.
#include "common.h"
#include "wino_plan.h"

namespace {

constexpr int WTCO = 64;      // output channels per workgroup
constexpr int WTT = 32;       // 2x4 output tiles per workgroup
constexpr int WCK = 8;        // input channels per chunk
constexpr int NSL = 24;       // Winograd slices (4 transformed rows x 6 transformed columns)
constexpr int WINO_PERSIST = 256;     // one persistent workgroup per CU

struct WinoArgs {
    const float* x;
    const float* u;       // packed transformed weights
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    int S, Cin, Cout, CinP, CoutP;
    // tail splitting (small maps): tiles [0, nfull) are computed whole (by the persistent workgroups); workgroup
    // npers + j*tail_split + p computes the p-th K range of tile nfull + j and leaves a raw partial output in ws
    // (wino_fixup_kernel)
    int nfull, tail_split;
    float* ws;
    int npers;            // persistent workgroups = min(nfull, WINO_PERSIST)
};

template <int LOGW, int MODE>
struct WGeo {
    static constexpr int W = 1 << LOGW, H = W, HW = W * H;
    static constexpr int SH = MODE == 2 ? H / 2 : H, SW = MODE == 2 ? W / 2 : W;   // source size
    static constexpr int TWC = W / 4;                // tiles per output row
    static constexpr int THR = H / 2;                // tile rows per image
    static constexpr int IPG = TWC * THR >= WTT ? 1 : WTT / (TWC * THR);   // images per workgroup (4 on 8x8 maps)
    static constexpr int TR = IPG == 1 ? WTT / TWC : THR;               // tile rows per image in a workgroup
    static constexpr int WPI = IPG == 1 ? THR / TR : 1;                 // workgroups per image (group)
    static constexpr int RPI = 2 * TR + 2;                              // haloed patch rows per image
    static constexpr int PH = IPG * RPI;
    // patch row: idx 3 = left halo, 4.. = pixels, 4+W = right halo.  With several images per workgroup the
    // row is W+4 wide and the right halo aliases the (never written, zero) idx 0 of the next row.
    // Stride: 16 consecutive tiles of a b128 lane group span 16/TWC tile rows = patch rows 2 apart; their 16-byte
    // reads must land on disjoint bank sets: 2*PW = 16*TWC mod 64 dwords (W = 32: 48 instead of 40; 64 and 16: W + 8).
    static constexpr int PW = IPG == 1 ? (W == 32 ? 48 : W + 8) : W + 4;
    static constexpr int PS = PH * PW + (IPG == 1 ? 0 : 8);
    static constexpr int Q = W / 4;
    static_assert(TR >= 1 && WPI >= 1 && IPG * TR * TWC == WTT, "unsupported map size for the 32-tile workgroup");
    static __device__ __forceinline__ int t_img(int tl) { return tl / (TWC * TR); }
    static __device__ __forceinline__ int t_row(int tl) { return (tl / TWC) % TR; }
    static __device__ __forceinline__ int t_col(int tl) { return tl % TWC; }
    static __device__ __forceinline__ int g_view(int wg) { return IPG == 1 ? wg / WPI : wg * IPG; }
    static __device__ __forceinline__ int g_row(int wg) { return IPG == 1 ? (wg % WPI) * 2 * TR : 0; }
    static int groups(int S) { return IPG == 1 ? S * WPI : (S + IPG - 1) / IPG; }
};

#ifdef VF_STAMPS
__device__ unsigned long long g_stamps[16];     // debug build only: shader-clock sums of prologue / chunk loop / epilogue, tiles
#endif
template <int LOGW, int MODE>
__global__ __launch_bounds__(512, 2) void wino_conv_kernel(WinoArgs a) {
    using G = WGeo<LOGW, MODE>;
    constexpr int NT_ = 512;
    constexpr int NX4 = WCK * G::PH * G::Q;
    constexpr int NXR = (NX4 + NT_ - 1) / NT_;           // 1-2 per thread (tail predicated)
    constexpr int UCH = NSL * WTCO * WCK;                // floats of one (co tile, chunk) block of U
    constexpr int VSZ = NSL * WCK * WTT;
    // (each raw-row buffer ends in a 512-byte scratch row: the LDS slot of an element that has no place in the patch -- a row
    // outside the image, the unused tail of the per-thread element list -- so that every store of the loop is unconditional)
    constexpr int PSZ = WCK * G::PS + 128;
    constexpr int XDUMMY = WCK * G::PS;
    constexpr int XCH = 2 * 4 * 64 * 64;                 // epilogue exchange: [cw][row wave][value 64][lane 64]
    constexpr int LDSF = 2 * PSZ + (2 * VSZ > XCH ? 2 * VSZ : XCH);
    static_assert(NXR <= 2, "raw-row staging assumes at most two float4 per thread");
    static_assert(LDSF * 4 <= 160 * 1024, "LDS budget");

    // V and the raw rows are double buffered so that ONE barrier per chunk suffices: while the MFMAs of chunk c read
    // V[c&1], the waves also store the raw rows of chunk c+2 and transform the rows of chunk c+1 into V[(c+1)&1].
    // The raw-row buffers come first: their zero halo columns are written once per workgroup and must survive the
    // epilogue's exchange buffer, which overlays the V buffers and the space behind them.
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    float* const Pl = lds;
    float* const Vl = lds + 2 * PSZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, wa = wid >> 1;               // channel half, transformed row of this wave
    const int li = lane & 31, lh = lane >> 5;
    const int ncot = a.CoutP / WTCO;
    const bool partial = (int)blockIdx.x >= a.npers;
    const int tail_id = partial ? (int)blockIdx.x - a.npers : 0;
    int c0 = 0, nch = a.CinP / WCK;                     // this workgroup's chunk range [c0, c0 + nch)
    if (partial) {
        const int per = (nch + a.tail_split - 1) / a.tail_split;
        c0 = (tail_id % a.tail_split) * per;
        nch = max(0, min(nch - c0, per));
    }
    const int clast = max(nch - 1, 0);

    // per-thread staging layout of the raw input rows: element e = tid + 512 i of the patch -> (channel, patch row,
    // float4 column); only what the chunk loop touches is kept in registers
    int xlo[2], xci[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = min(tid + i * NT_, NX4 - 1);
        const int t1 = e / G::Q;
        xci[i] = t1 / G::PH;
        xlo[i] = xci[i] * G::PS + (t1 % G::PH) * G::PW + 4 * (e % G::Q) + 4;
    }

    struct Tile {                                        // what changes from tile to tile and the chunk loop needs
        const char* ubase;                               // workgroup-uniform: U block of (co tile, first chunk)
        const char* xbase;                               // workgroup-uniform: first view of the tile
        int xgo0, xgo1;                                  // this thread's source byte offsets at channel xci (always a valid address)
        int xl0, xl1;                                    // ... and their LDS slots (the scratch row for elements outside the image)
        bool z0, z1;                                     // the element's own slot must hold zeros (row outside the image)
    };
    auto tile_pos = [&](unsigned logical, int& s_, int& r0_, int& cot_) {
        cot_ = logical % ncot;
        const int wg = logical / ncot;
        s_ = G::g_view(wg);                             // first (for 8x8 maps: of four) view of this tile
        r0_ = G::g_row(wg);                             // first output row
    };
    auto make_tile = [&](unsigned logical) -> Tile {
        Tile t;
        int ts, tr0, cot;
        tile_pos(logical, ts, tr0, cot);
        t.ubase = reinterpret_cast<const char*>(a.u + ((size_t)cot * (a.CinP / WCK) + c0) * UCH);
        t.xbase = reinterpret_cast<const char*>(a.x + (size_t)ts * a.Cin * (G::SH * G::SW));
        // Round 4: every load and LDS store of the chunk loop is UNCONDITIONAL (a load or store inside a branch costs the
        // compiler its count of the requests in flight; measured on the F(4x4) kernel).  An element outside the
        // image (or past the last view / the element list) reads the clamped row inside and lands in the scratch row.
        int go[2], xl[2];
        bool zz[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = min(tid + i * NT_, NX4 - 1);
            const int q = e % G::Q, t1 = e / G::Q;
            const int pr = t1 % G::PH;
            const int img = pr / G::RPI;
            const int uy = tr0 + pr % G::RPI - 1;
            const bool in_list = i < NXR && tid + i * NT_ < NX4;
            const bool ok = in_list && ts + img < a.S && uy >= 0 && uy < G::H;
            const int uyc = min(max(uy, 0), G::H - 1), imgc = min(img, max(a.S - 1 - ts, 0));
            go[i] = 4 * ((imgc * a.Cin + xci[i]) * (G::SH * G::SW) + (MODE == 0 ? uyc * G::SW + 4 * q : (uyc >> 1) * G::SW + 2 * q));
            xl[i] = ok ? xlo[i] : XDUMMY + 4 * (tid & 31);
            zz[i] = in_list && !ok;
        }
        t.xl0 = xl[0]; t.xl1 = xl[1]; t.z0 = zz[0]; t.z1 = zz[1];
        t.xgo0 = go[0]; t.xgo1 = go[1];
        return t;
    };
    // Under fp32 MFMAs every VALU instruction costs its full issue time (tools/mfma_valu.hip: VALU and the matrix pipe
    // do not overlap on a SIMD), so the loads of the chunk loop carry as little vector arithmetic as possible: a
    // uniform base in SGPRs + one 32-bit byte offset per thread (the saddr form of global_load).  A raw row outside
    // the image is not loaded at all: its LDS slot is zeroed once per tile (prologue) and the chunk loop's load /
    // store of that slot are masked off.  Channels beyond Cin (last chunk of a Cin that is no multiple of 8) read the
    // clamped last channel -- finite values whose packed weights are zero:
    //   min(8 c + xci, Cin - 1) = min(8 c, Cin - 1 - xci) + xci,  the "+ xci" part folded into the tile's offset.
    const int xlim[2] = {a.Cin - 1 - xci[0], a.Cin - 1 - xci[1]};
    unsigned uoffb = 4u * (unsigned)(((wa * 6) * WTCO + cw * 32 + li) * WCK + 4 * lh);    // this wave's fragment of U:
                                                                          // slices 6 wa .., rows cw*32 + li, k half lh
    auto fetch_x = [&](const Tile& t, int i, int c, float4& v) {
        const int go = i == 0 ? t.xgo0 : t.xgo1;
        const unsigned off = (unsigned)(min((c0 + c) * WCK, xlim[i]) * (4 * G::SH * G::SW) + go);
        const char* p = t.xbase + off;
        if (MODE == 0) v = *reinterpret_cast<const float4*>(p);
        else { const float2 h = *reinterpret_cast<const float2*>(p); v = make_float4(h.x, h.x, h.y, h.y); }
    };

    unsigned lin = partial ? 0u : blockIdx.x;           // linear id of the current whole tile
    const unsigned tail_logical = (unsigned)(a.nfull + tail_id / a.tail_split);
    auto logical_of = [&](unsigned l) -> unsigned { return partial ? tail_logical : xcd_remap(l, a.nfull); };
    Tile cur = make_tile(logical_of(lin));

    // named registers + macros (not arrays behind lambdas: those end up in scratch memory)
    f32x4 ur0, ur1, ur2, ur3, ur4, ur5;                // U fragments of the six slices, current chunk
    float4 xr0 = make_float4(0.f, 0.f, 0.f, 0.f), xr1 = xr0;
    // (the opaque copy keeps the uniform part of the address in an SGPR pair of its own: global_load ..., v_off, s[base]
    // offset:imm -- otherwise the six loads share one 64-bit VGPR base and pay v_add_co / v_addc pairs)
#define VF_ULOAD(T, B, C)                                                                               \
    {                                                                                                   \
        const char* ub_ = (T).ubase + ((size_t)(C) * UCH + ((B) & ~1) * (WTCO * WCK)) * 4;              \
        asm("" : "+s"(ub_), "+v"(uoffb));    /* (the offset too: its zero-extension must stay in this block) */ \
        ur##B = *(const __attribute__((address_space(1))) f32x4*)(                                      \
            (const __attribute__((address_space(1))) char*)ub_ + uoffb + ((B) & 1) * (WTCO * WCK * 4)); \
    }
#define VF_ULOAD_ALL(T, C) { VF_ULOAD(T, 0, C); VF_ULOAD(T, 1, C); VF_ULOAD(T, 2, C); VF_ULOAD(T, 3, C); VF_ULOAD(T, 4, C); VF_ULOAD(T, 5, C); }
#define VF_XLOAD(T, C, R0, R1) { fetch_x((T), 0, (C), R0); if (NXR > 1) fetch_x((T), 1, (C), R1); }
#define VF_XSTORE(T, BUF, R0, R1)                                                                       \
    {                                                                                                   \
        *reinterpret_cast<float4*>(Pl + (BUF) * PSZ + (T).xl0) = R0;                                     \
        if (NXR > 1) *reinterpret_cast<float4*>(Pl + (BUF) * PSZ + (T).xl1) = R1;                        \
    }
    // rows of the tile that lie outside the image (or past the last view): zero in both raw-row buffers
#define VF_XZERO(T)                                                                                     \
    {                                                                                                   \
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);                                              \
        if ((T).z0) {                                                                                   \
            *reinterpret_cast<float4*>(Pl + xlo[0]) = z4; *reinterpret_cast<float4*>(Pl + PSZ + xlo[0]) = z4; \
        }                                                                                               \
        if (NXR > 1 && (T).z1) {                                                                        \
            *reinterpret_cast<float4*>(Pl + xlo[1]) = z4; *reinterpret_cast<float4*>(Pl + PSZ + xlo[1]) = z4; \
        }                                                                                               \
    }

    // input transform duty: channel tci, tile ttl, half th (B2^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]).
    //   half 0 reads window rows (A,B,C) = (0,1,2) and writes transformed rows 0 = A - C and 1 = B + C;
    //   half 1 reads window rows (A,B,C) = (3,2,1) and writes transformed rows 3 and 2 = B - C -- and, with the SAME
    //   instruction as half 0, A - C = -(row 3): the packed weights of transformed row 3 carry the opposite sign
    //   (wino_pack_group), the products are bit-identical.
    // The rows are combined FIRST (on the raw 6-wide rows), the B4^T column transform then runs on two rows instead of
    // three, and both use packed fp32 (v_pk_add_f32 / v_pk_fma_f32 on the aligned pairs (d1,d2), (d3,d4)): 24 vector
    // instructions per thread and chunk instead of 66.
    // Tile of this lane: ds_read_b128 serves a wave in four groups of 16 lanes -- {0-3,12-15,20-27}, {4-11,16-19,28-31}
    // and the same +32 (MI355X_MICROARCH.md, LDS table) -- and a window read is conflict-free only if the 16 lanes of a
    // group cover 64 distinct banks.  The lanes of a group therefore take 16 CONSECUTIVE tiles (one tile row of a 64-wide
    // map, two of a 32-wide one, ...), whose 16-byte reads tile the banks exactly (the patch row stride is chosen
    // accordingly: WGeo::PW).
    const int l5 = tid & 31;
    const bool gB = (l5 >= 4 && l5 < 12) || (l5 >= 16 && l5 < 20) || l5 >= 28;
    const int gpos = gB ? (l5 < 12 ? l5 - 4 : (l5 < 20 ? l5 - 8 : l5 - 16)) : (l5 < 4 ? l5 : (l5 < 16 ? l5 - 8 : l5 - 12));
    const int tci = tid >> 6, ttl = (gB ? 16 : 0) + gpos, th = (tid >> 5) & 1;
    const int wpo = tci * G::PS + (G::t_img(ttl) * G::RPI + 2 * G::t_row(ttl)) * G::PW + 4 * G::t_col(ttl);
    const int wpoA = wpo + (th ? 3 : 0) * G::PW, wpoB = wpo + (th ? 2 : 1) * G::PW, wpoC = wpo + (th ? 1 : 2) * G::PW;
    // V[ci & 3][slice 24][ci >> 2][tile 32]: the MFMA lane (tile li, k half lh) reads channel 4 lh + e of slice s at
    // dword (e * 24 + s) * 64 + lane -- every fragment a multiple of 64 dwords from one per-lane base
    // (ds_read2st64_b32 with immediate offsets, no address arithmetic in the loop); a transform thread writes its
    // channel's 12 values the same way.
    const int wvb = ((tci & 3) * NSL) * 64 + (tci >> 2) * WTT + ttl;
    const int wvx = wvb + (th ? 18 : 0) * 64;                       // V slot of output X (transformed row 0 / 3), slice 0
    const int wvy = wvb + (th ? 12 : 6) * 64;                       // V slot of output Y (transformed row 1 / 2)
    const f32x2 sgy = th ? (f32x2){-1.f, -1.f} : (f32x2){1.f, 1.f};
    f32x2 xj0, x12, x34, x5j, yj0, y12, y34, y5j;        // combined rows X, Y: [junk, d0], [d1, d2], [d3, d4], [d5, junk]
    float cX[6], cY[6];                                  // their column transforms
    f32x4 qa0, qa1, qa2, qb0, qb1, qb2, qc0, qc1, qc2;   // raw window rows in flight (read in one MFMA gap, used in a later one)
    // one window row (cols 4t-1 .. 4t+4 = patch idx 4t+3 .. 4t+8): three aligned 16-byte reads, conflict-free
#define VF_WREAD(WPO, BUF, Q0, Q1, Q2)                                                   \
    {                                                                                    \
        const float* p_ = Pl + (BUF) * PSZ + (WPO);                                      \
        Q0 = *reinterpret_cast<const f32x4*>(p_);                                        \
        Q1 = *reinterpret_cast<const f32x4*>(p_ + 4);                                    \
        Q2 = *reinterpret_cast<const f32x4*>(p_ + 8);                                    \
    }
    auto win_rows = [&]() {
        // (only .w of the first and .x of the last read are needed; left alone the compiler narrows the 16-byte reads
        // to 8 / 4-byte ones, whose 4-dword lane stride is a bank conflict -- a ds_read_b128 at that stride is
        // conflict-free.  The opaque use keeps all four components, and with them the wide read, alive.)
        asm volatile("" : "+v"(qa0), "+v"(qa2), "+v"(qb0), "+v"(qb2), "+v"(qc0), "+v"(qc2));
        xj0 = pk_sub(qa0.zw, qc0.zw); x12 = pk_sub(qa1.xy, qc1.xy); x34 = pk_sub(qa1.zw, qc1.zw); x5j = pk_sub(qa2.xy, qc2.xy);
        yj0 = pk_fma(qc0.zw, sgy, qb0.zw); y12 = pk_fma(qc1.xy, sgy, qb1.xy); y34 = pk_fma(qc1.zw, sgy, qb1.zw);
        y5j = pk_fma(qc2.xy, sgy, qb2.xy);
    };
    // column transform B4^T of a combined row d0..d5 (4 packed + 4 scalar instructions):
    //   c0 = 4 d0 + (d4 - 5 d2)    c1 = (d4 - 4 d2) + (d3 - 4 d1)      c2 = (d4 - 4 d2) - (d3 - 4 d1)
    //   c5 = 4 d1 + (d5 - 5 d3)    c3 = (d4 - d2) + 2 (d3 - d1)        c4 = (d4 - d2) - 2 (d3 - d1)
    auto win_col = [&](const f32x2& vj0, const f32x2& v12, const f32x2& v34, const f32x2& v5j, float* c) {
        const f32x2 c12 = pk_hi_pm_lo(pk_nmul4_add(v12, v34));       // p = [d3 - 4 d1, d4 - 4 d2]
        const f32x2 c34 = pk_hi_pm_2lo(pk_sub(v34, v12));            // q = [d3 - d1,   d4 - d2]
        c[0] = __builtin_fmaf(4.f, vj0.y, __builtin_fmaf(-5.f, v12.y, v34.y));
        c[1] = c12.x; c[2] = c12.y; c[3] = c34.x; c[4] = c34.y;
        c[5] = __builtin_fmaf(4.f, v12.x, __builtin_fmaf(-5.f, v34.x, v5j.x));
    };
    auto win_write = [&](int b, int buf) {
        float* v = Vl + buf * VSZ;
        v[wvx + b * 64] = cX[b];
        v[wvy + b * 64] = cY[b];
    };
    const int voff = (wa * 6) * 64 + lane;

    // ---- first loads of the first tile: U(0), rows(0), rows(1) -- all issued together (one round trip)
    float4 yr0 = xr0, yr1 = xr0;
    VF_ULOAD_ALL(cur, 0);
    VF_XLOAD(cur, 0, xr0, xr1);
    VF_XLOAD(cur, min(1, clast), yr0, yr1);
    for (int i = tid; i < 2 * PSZ; i += NT_) Pl[i] = 0.f;    // halo columns stay zero in both buffers, for every tile

#ifdef VF_STAMPS
    unsigned long long st_pro = 0, st_loop = 0, st_epi = 0, st_tiles = 0, st_ea = 0, st_eb = 0, st_ec = 0, st_x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define VF_STAMP(V) const unsigned long long V = __builtin_amdgcn_s_memtime()
#else
#define VF_STAMP(V)
#endif
    // One slice = four MFMAs (K = 8) on accumulator B; side work goes BEHIND the slice's own MFMAs (both waves of a SIMD
    // run this code in phase: side work in front would idle the matrix pipe in both at once).  FIRST: the accumulator
    // starts from the literal 0 (no 96 v_mov per tile).
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
        VF_ULOAD(cur, B, min((C) + 1, clast));            /* this slice's U of the NEXT chunk, in place */ \
    }
    // B fragments of slice S_ of V buffer BUF_ -> register set D
#define VF_BFRAG(D, BUF_, S_)                                                                            \
    { _Pragma("unroll") for (int e = 0; e < 4; ++e) D[e] = Vl[(BUF_) * VSZ + voff + ((S_) + e * NSL) * 64]; }
    // One chunk, PAR = chunk parity = its V buffer (compile time: every LDS address of the loop is one per-thread base
    // + an immediate).  The B fragments rotate through three register sets; slice 0's are loaded by the previous chunk.
    //   slice 0: window rows of chunk C+1 read from P[PAR^1]        slice 1: their transform (24 vector instructions)
    //   slice 2: 12 values -> V[PAR^1]                              slice 3: raw rows of chunk C+2 -> P[PAR];
    //   ... the chunk's ONE barrier sits here, behind slice 3, with the fragments of slices 4 and 5 already in registers:
    //   slice 4: raw rows of chunk C+3 requested                    slice 5: slice-0 fragments of chunk C+1 from V[PAR^1]
    // so that no wave waits for LDS at a chunk boundary.  Hazards: V[PAR] is rewritten in slice 2 of chunk C+1 (behind
    // this barrier, all its reads are in front); P[PAR^1] is rewritten in slice 3 of chunk C+1, read in slice 0 of C.
    // The staging code of chunks C+1..C+3 runs unconditionally with the chunk index clamped to the last one (the final
    // iterations redo harmless loads / LDS writes that nobody reads).
#define VF_CHUNK(C, PAR, FIRST)                                                                          \
    {                                                                                                    \
        VF_SLICE(C, FIRST, 0, bfA, { VF_BFRAG(bfB, PAR, 1); VF_WREAD(wpoA, (PAR) ^ 1, qa0, qa1, qa2); }, \
                 VF_WREAD(wpoB, (PAR) ^ 1, qb0, qb1, qb2), VF_WREAD(wpoC, (PAR) ^ 1, qc0, qc1, qc2));    \
        VF_SLICE(C, FIRST, 1, bfB, { VF_BFRAG(bfA, PAR, 2); win_rows(); }, win_col(xj0, x12, x34, x5j, cX), \
                 win_col(yj0, y12, y34, y5j, cY));                                                       \
        VF_SLICE(C, FIRST, 2, bfA, { VF_BFRAG(bfB, PAR, 3); win_write(0, (PAR) ^ 1); win_write(1, (PAR) ^ 1); }, \
                 { win_write(2, (PAR) ^ 1); win_write(3, (PAR) ^ 1); },                                  \
                 { win_write(4, (PAR) ^ 1); win_write(5, (PAR) ^ 1); });                                 \
        VF_SLICE(C, FIRST, 3, bfB, { VF_BFRAG(bfA, PAR, 4); VF_BFRAG(bfC, PAR, 5); },                    \
                 VF_XSTORE(cur, PAR, xr0, xr1), (void)0);                                                \
        VF_LDS_BARRIER();                                                                                 \
        VF_SLICE(C, FIRST, 4, bfA, VF_XLOAD(cur, min((C) + 3, clast), xr0, xr1), (void)0, (void)0);      \
        VF_SLICE(C, FIRST, 5, bfC, VF_BFRAG(bfA, (PAR) ^ 1, 0), (void)0, (void)0);                       \
    }

    // Staging of a tile's first chunks: rows(0), rows(1) -> P[0], P[1] (rows(2) requested), then V(0) from P[0].
    // For the first tile of the workgroup it runs here; for every later tile it runs INSIDE the previous tile's
    // epilogue (the raw-row buffers are not part of the exchange area; V[0] is free once every wave has read its
    // exchanged values), where it overlaps with the output stores.
#define VF_STAGE_ROWS(T)                                                                                 \
    {                                                                                                    \
        VF_XZERO(T);                                                                                     \
        VF_XSTORE(T, 0, xr0, xr1);                                                                       \
        VF_XSTORE(T, 1, yr0, yr1);                                                                       \
        VF_XLOAD(T, min(2, clast), xr0, xr1);                                                            \
    }
#define VF_STAGE_V0()                                                                                    \
    {                                                                                                    \
        VF_WREAD(wpoA, 0, qa0, qa1, qa2); VF_WREAD(wpoB, 0, qb0, qb1, qb2); VF_WREAD(wpoC, 0, qc0, qc1, qc2); \
        win_rows(); win_col(xj0, x12, x34, x5j, cX); win_col(yj0, y12, y34, y5j, cY);                    \
        _Pragma("unroll") for (int b = 0; b < 6; ++b) win_write(b, 0);                                   \
    }
    __syncthreads();                                      // zero fill done
    VF_STAGE_ROWS(cur);
    __syncthreads();
    VF_STAGE_V0();
    __syncthreads();

    for (;;) {
        VF_STAMP(t_0);
        f32x16 acc[6];
        float bfA[4], bfB[4], bfC[4];
        VF_BFRAG(bfA, 0, 0);
        VF_STAMP(t_1);

        // chunk 0 apart (zero accumulators), then pairs (odd, even), then the odd one left over
        if (nch > 0) {
            VF_CHUNK(0, 0, 1);
        } else {
#pragma unroll
            for (int b = 0; b < 6; ++b) acc[b] = (f32x16){0};
        }
        {
            int c = 1;
            for (; c + 1 < nch; c += 2) {
                VF_CHUNK(c, 1, 0);
                VF_CHUNK(c + 1, 0, 0);
            }
            if (c < nch) VF_CHUNK(c, 1, 0);
        }

        VF_STAMP(t_2);
        const unsigned lin_next = lin + WINO_PERSIST;
        const bool has_next = !partial && lin_next < (unsigned)a.nfull;      // workgroup-uniform
        const unsigned logical_cur = logical_of(lin);

        // ---- output transform, columns: this wave holds transformed row `wa`, M[b] = acc[b]:  t = A4^T M,
        //   A4^T = [[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]]
        // (packed fp32 on the register pairs (r, r+1) of the accumulators: 10 instructions per pair)
        f32x2 part[8][4];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 m0 = (f32x2){acc[0][2 * k], acc[0][2 * k + 1]};
            const f32x2 m1 = (f32x2){acc[1][2 * k], acc[1][2 * k + 1]};
            const f32x2 m2 = (f32x2){acc[2][2 * k], acc[2][2 * k + 1]};
            const f32x2 m3 = (f32x2){acc[3][2 * k], acc[3][2 * k + 1]};
            const f32x2 m4 = (f32x2){acc[4][2 * k], acc[4][2 * k + 1]};
            const f32x2 m5 = (f32x2){acc[5][2 * k], acc[5][2 * k + 1]};
            const f32x2 s12 = pk_add(m1, m2), d12 = pk_sub(m1, m2), s34 = pk_add(m3, m4), d34 = pk_sub(m3, m4);
            part[k][0] = pk_add(pk_add(m0, s12), s34);
            part[k][1] = pk_fmak<2>(d34, d12);
            part[k][2] = pk_fmak<4>(s34, s12);
            part[k][3] = pk_add(pk_fmak<8>(d34, d12), m5);
        }
        // ---- the next whole tile of this (persistent) workgroup: its first loads go out now (the accumulators are
        // dead, so the registers are free) and land under the rest of the epilogue.  Issued unconditionally -- the last
        // tile re-reads its own first chunks, which nobody uses -- so that nothing is conditionally carried through
        // the chunk loop.
        const Tile nx = make_tile(has_next ? logical_of(lin_next) : logical_cur);
        VF_ULOAD_ALL(nx, 0);
        VF_XLOAD(nx, 0, xr0, xr1);
        VF_XLOAD(nx, min(1, clast), yr0, yr1);

        // The epilogue's per-lane index arithmetic is the same for every tile; left to itself the compiler hoists all of
        // it out of the tile loop and then spills it across the chunk loop.  An opaque copy of the lane id keeps it here.
        VF_STAMP(t_a);
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, lh_e = lane_e >> 5;
        // ---- rows: Y = A2^T T, A2^T = [[1,1,1,0],[0,1,-1,-1]].  Every wave publishes its 16 x 4 values, then finishes
        // the four accumulator rows r = 4 wa .. 4 wa + 3 (channels co0 + cw*32 + 8 wa + rr + 4 lh) of all four row waves.
        float* xch = Vl + (size_t)(cw * 4) * (64 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) xch[(size_t)wa * (64 * 64) + (r * 4 + j) * 64 + lane_e] = part[r >> 1][j][r & 1];
        VF_LDS_BARRIER();
        VF_STAMP(t_b);
        float o0[4][4], o1[4][4];                        // [rr][column]: output rows 0 / 1 of the 2x4 tile
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float* q = xch + ((4 * wa + rr) * 4 + j) * 64 + lane_e;
                const float t0 = q[0], t1 = q[64 * 64], t2 = q[2 * 64 * 64], t3 = q[3 * 64 * 64];
                o0[rr][j] = t0 + t1 + t2;
                o1[rr][j] = t1 - t2 - t3;
            }
        VF_STAMP(t_c);
        int s, r0, cot_;
        tile_pos(logical_cur, s, r0, cot_);
        const int co0 = cot_ * WTCO;
        const int tl = li_e;
        const int cob = co0 + cw * 32 + 8 * wa + 4 * lh_e;                  // + rr
        const int sv = s + G::t_img(tl);
        const int orow = r0 + 2 * G::t_row(tl), ocol = 4 * G::t_col(tl);
        const bool ok_out = !partial && sv < a.S;
        // every epilogue operand is fetched BEFORE the first store (loads and stores retire through one in-order
        // counter); one uniform branch per operand kind with its loads back to back
        float eb[4], ev[4];
        float4 er[4][2];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            eb[rr] = ev[rr] = 0.f;
            er[rr][0] = er[rr][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (ok_out) {
            if (a.bias) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) eb[rr] = a.bias[min(cob + rr, a.Cout - 1)];
            }
            if (a.vbias) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) ev[rr] = a.vbias[(size_t)sv * a.Cout + min(cob + rr, a.Cout - 1)];
            }
            if (a.res) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const size_t o = ((size_t)sv * a.Cout + min(cob + rr, a.Cout - 1)) * G::HW + (size_t)orow * G::W + ocol;
                    er[rr][0] = *reinterpret_cast<const float4*>(a.res + o);
                    er[rr][1] = *reinterpret_cast<const float4*>(a.res + o + G::W);
                }
            }
        }
        // ---- next tile, staged under this tile's stores: raw rows now (P is outside the exchange area), V(0) once
        // every wave has taken its values out of the exchange area
        VF_STAMP(t_d);
        if (has_next) VF_STAGE_ROWS(nx);
        VF_STAMP(t_e);
        VF_LDS_BARRIER();
        VF_STAMP(t_f);
        if (partial) {                                       // raw partial tile: ws[tail_id][co 64][tile 32][2x4]
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = cw * 32 + 8 * wa + 4 * lh_e + rr;
                float* w8 = a.ws + (((size_t)tail_id * WTCO + col) * WTT + tl) * 8;
                *reinterpret_cast<float4*>(w8) = make_float4(o0[rr][0], o0[rr][1], o0[rr][2], o0[rr][3]);
                *reinterpret_cast<float4*>(w8 + 4) = make_float4(o1[rr][0], o1[rr][1], o1[rr][2], o1[rr][3]);
            }
        } else if (ok_out) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (cob + rr >= a.Cout) continue;
                const float bb = eb[rr] + ev[rr];
                const size_t o = ((size_t)sv * a.Cout + cob + rr) * G::HW + (size_t)orow * G::W + ocol;
                *reinterpret_cast<float4*>(a.y + o) =
                    make_float4(o0[rr][0] + bb + er[rr][0].x, o0[rr][1] + bb + er[rr][0].y,
                                o0[rr][2] + bb + er[rr][0].z, o0[rr][3] + bb + er[rr][0].w);
                *reinterpret_cast<float4*>(a.y + o + G::W) =
                    make_float4(o1[rr][0] + bb + er[rr][1].x, o1[rr][1] + bb + er[rr][1].y,
                                o1[rr][2] + bb + er[rr][1].z, o1[rr][3] + bb + er[rr][1].w);
            }
        }
#ifdef VF_STAMPS
        {
            const unsigned long long t_3 = __builtin_amdgcn_s_memtime();
            st_pro += t_1 - t_0; st_loop += t_2 - t_1; st_epi += t_3 - t_2; st_tiles += 1; st_ea += t_a - t_2; st_eb += t_b - t_a; st_ec += t_c - t_b;
        }
#endif
        if (!has_next) break;
        VF_STAMP(t_g);
        VF_STAGE_V0();
        VF_STAMP(t_h);
        VF_LDS_BARRIER();                                  // V(0) of the next tile complete
#ifdef VF_STAMPS
        {
            const unsigned long long t_i = __builtin_amdgcn_s_memtime();
            st_x[0] += t_d - t_c; st_x[1] += t_e - t_d; st_x[2] += t_f - t_e; st_x[3] += t_g - t_f; st_x[4] += t_h - t_g; st_x[5] += t_i - t_h; st_x[6] += 1;
        }
#endif
        cur = nx;
        lin = lin_next;
    }
#ifdef VF_STAMPS
    if (tid == 0 && !partial) {
        atomicAdd(&g_stamps[0], st_pro); atomicAdd(&g_stamps[1], st_loop); atomicAdd(&g_stamps[2], st_epi);
        atomicAdd(&g_stamps[3], st_tiles); atomicAdd(&g_stamps[4], st_ea); atomicAdd(&g_stamps[5], st_eb); atomicAdd(&g_stamps[6], st_ec);
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stamps[8 + i], st_x[i]);
    }
#endif
#undef VF_STAMP
#undef VF_ULOAD
#undef VF_ULOAD_ALL
#undef VF_XLOAD
#undef VF_XSTORE
#undef VF_XZERO
#undef VF_STAGE_ROWS
#undef VF_STAGE_V0
#undef VF_SLICE
#undef VF_BFRAG
#undef VF_CHUNK
#undef VF_WREAD
}

// OIHW -> transformed + packed forward  U[co tile][ci chunk][slice][co 64][ci 8] = (G2 w G4^T)_slice, slice = 6 a + b,
//        and backward (dgrad)          [ci tile][co chunk][slice][ci 64][co 8] of the 180-degree-rotated kernel.
// One 512-thread workgroup per (tile, chunk) group = 12288 outputs: thread (m, k8) reads the nine taps of one
// (co, ci) pair once (36 contiguous bytes) and writes its 24 slices, each slice a contiguous 2 KB line of the
// workgroup.  Groups [0, nf/12288) are the forward pack, the rest the backward pack.  Evaluated in double.
__device__ __forceinline__ void wino_pack_group(const float* __restrict__ w, float* __restrict__ uf,
                                                float* __restrict__ ub, int Cout, int Cin, size_t nf, size_t nb,
                                                size_t group) {
    constexpr int GSZ = NSL * WTCO * WCK;
    const size_t ngf = nf / GSZ;
    const bool bwd = group >= ngf;
    if (bwd) {
        group -= ngf;
        if (group >= nb / GSZ || !ub) return;
    }
    const int M = bwd ? Cin : Cout, K = bwd ? Cout : Cin;
    const int nchunk = (K + WCK - 1) / WCK;
    const int chunk = group % nchunk, mt = group / nchunk;
    const int t = threadIdx.x;
    const int m = t >> 3, k8 = t & 7;
    const int mm = mt * 64 + m, kk = chunk * WCK + k8;
    double g[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = 0.0;
    if (mm < M && kk < K) {
        const int co = bwd ? kk : mm, ci = bwd ? mm : kk;
        const float* p = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = (double)(bwd ? p[8 - i] : p[i]);
    }
    // rows: G2 = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]] applied to the kernel rows
    double tq[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double h = 0.5 * (g[q] + g[6 + q]), e = 0.5 * g[3 + q];
        tq[0][q] = g[q];
        tq[1][q] = h + e;
        tq[2][q] = h - e;
        tq[3][q] = g[6 + q];
    }
    // columns: G4 = [[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]]
    float* out = (bwd ? ub : uf) + group * (size_t)GSZ + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // (transformed row 3 is stored NEGATED: the conv kernel's input transform produces -(row 3), see there)
        const double x0 = i == 3 ? -tq[i][0] : tq[i][0], x1 = i == 3 ? -tq[i][1] : tq[i][1], x2 = i == 3 ? -tq[i][2] : tq[i][2];
        const double a6 = -(x0 + x2) / 6.0, b6 = x1 / 6.0;
        const double a24 = x0 / 24.0 + x2 / 6.0, b12 = x1 / 12.0;
        out[(6 * i + 0) * (WTCO * WCK)] = (float)(x0 / 4.0);
        out[(6 * i + 1) * (WTCO * WCK)] = (float)(a6 - b6);
        out[(6 * i + 2) * (WTCO * WCK)] = (float)(a6 + b6);
        out[(6 * i + 3) * (WTCO * WCK)] = (float)(a24 + b12);
        out[(6 * i + 4) * (WTCO * WCK)] = (float)(a24 - b12);
        out[(6 * i + 5) * (WTCO * WCK)] = (float)x2;
    }
}

__global__ __launch_bounds__(512) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                        float* __restrict__ ub, int Cout, int Cin, size_t nf,
                                                        size_t nb) {
    wino_pack_group(w, uf, ub, Cout, Cin, nf, nb, blockIdx.x);
}

constexpr int PACK_BLOCKS = NSL * WTCO * WCK / 256;     // 256-output units per pack group (48)

struct WPackDesc {
    const float* w;
    float* uf;
    float* ub;
    long long Cout, Cin, nf, nb, first_block;         // first_block in units of 256 outputs (48 per group)
};
__global__ __launch_bounds__(512) void wino_pack_multi_kernel(const WPackDesc* __restrict__ desc, int nlayers) {
    const long long vb = (long long)blockIdx.x * PACK_BLOCKS;
    int lo = 0, hi = nlayers;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= vb) lo = mid; else hi = mid;
    }
    const WPackDesc d = desc[lo];
    wino_pack_group(d.w, d.uf, d.ub, (int)d.Cout, (int)d.Cin, (size_t)d.nf, (size_t)d.nb,
                    (size_t)((vb - d.first_block) / PACK_BLOCKS));
}

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// Sums the K-range partials of the tail tiles in a fixed order and applies the epilogue
// (bias + per-view bias + residual).  One thread per (tail tile, co, 2x4 tile).
template <int LOGW>
__global__ __launch_bounds__(256) void wino_fixup_kernel(WinoArgs a, int ntail) {
    using G = WGeo<LOGW, 0>;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int tl = idx % WTT;
    const int col = (idx / WTT) % WTCO;
    const int j = idx / (WTT * WTCO);
    if (j >= ntail) return;
    const int ncot = a.CoutP / WTCO;
    const int logical = a.nfull + j;
    const int cot = logical % ncot, wg = logical / ncot;
    const int s = G::g_view(wg) + G::t_img(tl), r0 = G::g_row(wg);
    const int co = cot * WTCO + col;
    if (s >= a.S || co >= a.Cout) return;
    // every operand is requested before the first add (tail_split <= 8 partials on clamped indices, then bias and
    // residual): a plain accumulate loop pays one memory round trip per partial.  Fixed summation order.
    const int orow = r0 + 2 * G::t_row(tl), ocol = 4 * G::t_col(tl);
    const size_t o = ((size_t)s * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
    float4 t0[8], t1[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const float* w8 = a.ws + ((((size_t)j * a.tail_split + min(p, a.tail_split - 1)) * WTCO + col) * WTT + tl) * 8;
        t0[p] = *reinterpret_cast<const float4*>(w8);
        t1[p] = *reinterpret_cast<const float4*>(w8 + 4);
    }
    float b = 0.f;
    if (a.bias) b += a.bias[co];
    if (a.vbias) b += a.vbias[(size_t)s * a.Cout + co];
    float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
    if (a.res) {
        q0 = *reinterpret_cast<const float4*>(a.res + o);
        q1 = *reinterpret_cast<const float4*>(a.res + o + G::W);
    }
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        if (p < a.tail_split) {
            v0.x += t0[p].x; v0.y += t0[p].y; v0.z += t0[p].z; v0.w += t0[p].w;
            v1.x += t1[p].x; v1.y += t1[p].y; v1.z += t1[p].z; v1.w += t1[p].w;
        }
    }
    v0.x += b; v0.y += b; v0.z += b; v0.w += b;
    v1.x += b; v1.y += b; v1.z += b; v1.w += b;
    v0.x += q0.x; v0.y += q0.y; v0.z += q0.z; v0.w += q0.w;
    v1.x += q1.x; v1.y += q1.y; v1.z += q1.z; v1.w += q1.w;
    *reinterpret_cast<float4*>(a.y + o) = v0;
    *reinterpret_cast<float4*>(a.y + o + G::W) = v1;
}

// Fix-up + GroupNorm(+Swish) in ONE launch (round 5; the sampler at a few views, where EVERY tile of the launch is a K-split
// tail tile): one workgroup per (view, group) sums the K-range partials of its channels in the fix-up's fixed order, adds
// bias / per-view bias / residual, optionally stores the conv output and -- the whole group being in its registers --
// normalises it (two-pass mean / variance, as gn_fwd_kernel and conv_splitk_reduce_gn_kernel) into a_out.  Replaces
// wino_fixup_kernel + gn_fwd_kernel: one graph node less per 3x3 conv of the 64x64 / 32x32 levels at S = 2 ... 16.
// Element = one 2x4 output tile of one channel (8 floats); NV elements per thread, 512 threads.
template <int LOGW, int NV>
__global__ __launch_bounds__(512) void wino_fixup_gn_kernel(WinoArgs a, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ a_out,
                                                            int store_y, int cpg, float eps, int silu) {
    using G = WGeo<LOGW, 0>;
    static_assert(G::IPG == 1, "maps of at least 16x16");
    constexpr int NT = 512, TPV = G::HW / 8;             // 2x4 tiles per view
    __shared__ float red[NT / 64];
    const int ngroups = a.Cout / cpg;
    const int s = blockIdx.x / ngroups, g = blockIdx.x - s * ngroups;
    const int nel = cpg * TPV, ncot = a.CoutP / WTCO;
    float4 v0[NV], v1[NV];
    size_t oo[NV];
    float gam[NV], bet[NV], badd[NV];
    float4 t0[NV][4], t1[NV][4];
    const float* wsb[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {                       // every operand is requested before the first add
        const int e = min((int)threadIdx.x + i * NT, nel - 1);
        const int c = g * cpg + e / TPV, tv = e % TPV;
        const int wg = s * G::WPI + tv / WTT, tl = tv % WTT;
        const int j = wg * ncot + c / WTCO;              // (nfull == 0: the logical tile IS the tail index)
        wsb[i] = a.ws + (((size_t)j * a.tail_split * WTCO + c % WTCO) * WTT + tl) * 8;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float* w8 = wsb[i] + (size_t)min(p, a.tail_split - 1) * (WTCO * WTT * 8);
            t0[i][p] = *reinterpret_cast<const float4*>(w8);
            t1[i][p] = *reinterpret_cast<const float4*>(w8 + 4);
        }
        const int orow = G::g_row(wg) + 2 * G::t_row(tl), ocol = 4 * G::t_col(tl);
        oo[i] = ((size_t)s * a.Cout + c) * G::HW + (size_t)orow * G::W + ocol;
        float b = 0.f;
        if (a.bias) b += a.bias[c];
        if (a.vbias) b += a.vbias[(size_t)s * a.Cout + c];
        badd[i] = b;
        gam[i] = gamma[c];
        bet[i] = beta[c];
        v0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        v1[i] = v0[i];
        if (a.res) {
            v0[i] = *reinterpret_cast<const float4*>(a.res + oo[i]);
            v1[i] = *reinterpret_cast<const float4*>(a.res + oo[i] + G::W);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
#pragma unroll
        for (int p = 0; p < 4; ++p) {                    // fixed summation order, as wino_fixup_kernel
            if (p < a.tail_split) {
                q0.x += t0[i][p].x; q0.y += t0[i][p].y; q0.z += t0[i][p].z; q0.w += t0[i][p].w;
                q1.x += t1[i][p].x; q1.y += t1[i][p].y; q1.z += t1[i][p].z; q1.w += t1[i][p].w;
            }
        }
        if (a.tail_split > 4) {                          // partials 4 .. 7: a second batch of loads (deep-K layers only)
            float4 u0[4], u1[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float* w8 = wsb[i] + (size_t)min(4 + p, a.tail_split - 1) * (WTCO * WTT * 8);
                u0[p] = *reinterpret_cast<const float4*>(w8);
                u1[p] = *reinterpret_cast<const float4*>(w8 + 4);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (4 + p < a.tail_split) {
                    q0.x += u0[p].x; q0.y += u0[p].y; q0.z += u0[p].z; q0.w += u0[p].w;
                    q1.x += u1[p].x; q1.y += u1[p].y; q1.z += u1[p].z; q1.w += u1[p].w;
                }
            }
        }
        const float b = badd[i];
        q0.x += b; q0.y += b; q0.z += b; q0.w += b;
        q1.x += b; q1.y += b; q1.z += b; q1.w += b;
        v0[i] = make_float4(q0.x + v0[i].x, q0.y + v0[i].y, q0.z + v0[i].z, q0.w + v0[i].w);
        v1[i] = make_float4(q1.x + v1[i].x, q1.y + v1[i].y, q1.z + v1[i].z, q1.w + v1[i].w);
        const bool ok = (int)threadIdx.x + i * NT < nel;
        if (store_y && ok) {
            *reinterpret_cast<float4*>(a.y + oo[i]) = v0[i];
            *reinterpret_cast<float4*>(a.y + oo[i] + G::W) = v1[i];
        }
        if (!ok) { v0[i] = make_float4(0.f, 0.f, 0.f, 0.f); v1[i] = v0[i]; }
        sum += ((v0[i].x + v0[i].y) + (v0[i].z + v0[i].w)) + ((v1[i].x + v1[i].y) + (v1[i].z + v1[i].w));
    }
    const float inv_n = 1.0f / (float)(nel * 8);
    const float mean = block_sum<NT>(sum, red) * inv_n;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float d0 = v0[i].x - mean, d1 = v0[i].y - mean, d2 = v0[i].z - mean, d3 = v0[i].w - mean;
        const float d4 = v1[i].x - mean, d5 = v1[i].y - mean, d6 = v1[i].z - mean, d7 = v1[i].w - mean;
        const float q = ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3)) + ((d4 * d4 + d5 * d5) + (d6 * d6 + d7 * d7));
        sq += (int)threadIdx.x + i * NT < nel ? q : 0.f;
    }
    const float var = block_sum<NT>(sq, red) * inv_n;
    const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if ((int)threadIdx.x + i * NT < nel) {
            const float ga = gam[i] * rstd, be = bet[i] - mean * ga;
            float4 o0 = make_float4(v0[i].x * ga + be, v0[i].y * ga + be, v0[i].z * ga + be, v0[i].w * ga + be);
            float4 o1 = make_float4(v1[i].x * ga + be, v1[i].y * ga + be, v1[i].z * ga + be, v1[i].w * ga + be);
            if (silu) {
                o0.x = silu_f(o0.x); o0.y = silu_f(o0.y); o0.z = silu_f(o0.z); o0.w = silu_f(o0.w);
                o1.x = silu_f(o1.x); o1.y = silu_f(o1.y); o1.z = silu_f(o1.z); o1.w = silu_f(o1.w);
            }
            *reinterpret_cast<float4*>(a_out + oo[i]) = o0;
            *reinterpret_cast<float4*>(a_out + oo[i] + G::W) = o1;
        }
    }
}

// tiles of a launch: groups of 32 output tiles (256 pixels) x 64-channel tiles
inline int wino_groups(int S, int H, int W) {
    const int tiles = (H / 2) * (W / 4);
    return tiles >= WTT ? S * (tiles / WTT) : (S + WTT / tiles - 1) / (WTT / tiles);
}

struct WinoGn {                      // GroupNorm(+Swish) behind the conv, evaluated by the fix-up launch (or gamma = null)
    const float* gamma = nullptr;
    const float* beta = nullptr;
    float* a_out = nullptr;
    int groups = 0, silu = 0, store_y = 1;
    float eps = 0.f;
};

// Can the fix-up launch of this shape evaluate the GroupNorm?  Every tile a K-split tail tile, maps of
// at least 16x16 (one view per workgroup), a (view, group) of at most 2048 2x4 tiles.
inline bool wino_gn_fusable(int S, int Cin, int Cout, int H, int W, int groups) {
    if (W < 16 || groups <= 0 || Cout % groups != 0) return false;
    const int T = wino_groups(S, H, W) * (rup(Cout, WTCO) / WTCO);
    int nfull, split;
    wino_tail_plan(T, rup(Cin, WCK) / WCK, &nfull, &split);
    return nfull == 0 && split >= 2 && split <= 8 && (long)(Cout / groups) * (H * W / 8) <= 2048;
}

template <int LOGW, int MODE>
int launch_wino(WinoArgs a, size_t ws_floats, hipStream_t st, const WinoGn& gn = WinoGn()) {
    using G = WGeo<LOGW, MODE>;
    const int T = G::groups(a.S) * (a.CoutP / WTCO);
    wino_tail_plan(T, a.CinP / WCK, &a.nfull, &a.tail_split);
    const int ntail = T - a.nfull;
    if ((size_t)ntail * a.tail_split * WTCO * WTT * 8 > ws_floats || !a.ws) {   // no room: plain grid
        a.nfull = T;
        a.tail_split = 1;
    }
    const int nt = T - a.nfull;
    if (gn.gamma && !(a.nfull == 0 && a.tail_split >= 2)) return (int)hipErrorInvalidValue;     // (callers ask wino_gn_fusable first)
    a.npers = a.nfull < WINO_PERSIST ? a.nfull : WINO_PERSIST;
    hipLaunchKernelGGL((wino_conv_kernel<LOGW, MODE>), dim3(a.npers + nt * a.tail_split), dim3(512), 0, st, a);
    if constexpr (LOGW >= 4) {
        if (gn.gamma) {
            const int cpg = a.Cout / gn.groups;
            const long nel = (long)cpg * ((1 << (2 * LOGW)) / 8);
            const dim3 grid(a.S * gn.groups);
#define VF_FGN(NV) hipLaunchKernelGGL((wino_fixup_gn_kernel<LOGW, NV>), grid, dim3(512), 0, st, a, gn.gamma, gn.beta, gn.a_out, \
                                      gn.store_y, cpg, gn.eps, gn.silu)
            if (nel <= 512) VF_FGN(1);
            else if (nel <= 1024) VF_FGN(2);
            else VF_FGN(4);
#undef VF_FGN
            VF_RETURN_LAST_ERROR();
        }
    }
    if (nt > 0)
        hipLaunchKernelGGL((wino_fixup_kernel<LOGW>), dim3((nt * WTCO * WTT + 255) / 256), dim3(256), 0, st, a, nt);
    VF_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

int vf_wino_pack_sizes(int Cout, int Cin, long* fwd_floats, long* bwd_floats) {
    *fwd_floats = (long)NSL * rup(Cin, WCK) * rup(Cout, WTCO);
    *bwd_floats = (long)NSL * rup(Cout, WCK) * rup(Cin, WTCO);
    return 0;
}

int vf_wino_pack_weights(const float* w_oihw, float* u_fwd, float* u_bwd, int Cout, int Cin, void* stream) {
    const size_t nf = (size_t)NSL * rup(Cin, WCK) * rup(Cout, WTCO);
    const size_t nb = u_bwd ? (size_t)NSL * rup(Cout, WCK) * rup(Cin, WTCO) : 0;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((nf + nb) / (NSL * WTCO * WCK))), dim3(512), 0,
                       (hipStream_t)stream, w_oihw, u_fwd, u_bwd, Cout, Cin, nf, nb);
    VF_RETURN_LAST_ERROR();
}

// desc: device int64 [nlayers][8] rows {w, u_fwd, u_bwd, Cout, Cin, fwd_floats, bwd_floats, first_block}
int vf_wino_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream) {
    if (nlayers <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(wino_pack_multi_kernel, dim3((unsigned)(total_blocks / PACK_BLOCKS)), dim3(512), 0,
                       (hipStream_t)stream, (const WPackDesc*)desc, nlayers);
    VF_RETURN_LAST_ERROR();
}

// 1 if vf_wino_conv_fwd supports this (output) size / mode: 3x3 stride 1, H = W in {8, 16, 32, 64}, modes 0 / 2.
int vf_wino_supported(int H, int W, int mode) {
    return (H == W && (W == 8 || W == 16 || W == 32 || W == 64) && (mode == 0 || mode == 2)) ? 1 : 0;
}

// workspace floats vf_wino_conv_fwd wants for its split tail tiles (0 when the grid divides evenly)
long vf_wino_conv_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const int T = wino_groups(S, H, W) * (rup(Cout, WTCO) / WTCO);
    int nfull, split;
    wino_tail_plan(T, rup(Cin, WCK) / WCK, &nfull, &split);
    return (long)(T - nfull) * split * WTCO * WTT * 8;
}

// Expected CU fill (percent) of vf_wino_conv_fwd at this shape, and the tile count; hosts use it to choose between
// this path and the direct kernel.  (Pure occupancy figure: the tail is priced WITHOUT the per-part overhead that
// the launch's own plan uses, so the hosts' thresholds keep their meaning.)
int vf_wino_conv_fill_pct(int S, int Cin, int Cout, int H, int W, int* tiles_out) {
    const int T = wino_groups(S, H, W) * (rup(Cout, WTCO) / WTCO);
    const int nch = rup(Cin, WCK) / WCK;
    int nfull, split;
    wino_tail_plan(T, nch, &nfull, &split, 0.0);
    if (tiles_out) *tiles_out = T;
    if (T <= 0) return 0;
    const double time = (nfull + WINO_SLOTS - 1) / WINO_SLOTS + (T > nfull ? wino_tail_time(T - nfull, split, nch, 0.0) : 0.0);
    return (int)(100.0 * T / WINO_SLOTS / time);
}

// y = conv3x3(x) (+bias +view_bias +residual), pad 1, stride 1, via the fused nested Winograd F(2,3) x F(4,3).
// u_packed from vf_wino_pack_weights (forward pack for the conv, backward pack for its dgrad).
int vf_wino_conv_fwd(const float* x, const float* u_packed, const float* bias, const float* view_bias,
                     const float* residual, float* y, float* ws, long ws_floats, int S, int Cin, int Cout, int H,
                     int W, int mode, void* stream) {
    if (S <= 0) return 0;
    if (!vf_wino_supported(H, W, mode)) return (int)hipErrorInvalidValue;
    WinoArgs a;
    a.x = x; a.u = u_packed; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.S = S; a.Cin = Cin; a.Cout = Cout; a.CinP = rup(Cin, WCK); a.CoutP = rup(Cout, WTCO);
    a.ws = ws;
    hipStream_t st = (hipStream_t)stream;
    const size_t nws = ws ? (size_t)ws_floats : 0;
    if (W == 8) return mode == 0 ? launch_wino<3, 0>(a, nws, st) : launch_wino<3, 2>(a, nws, st);
    if (W == 16) return mode == 0 ? launch_wino<4, 0>(a, nws, st) : launch_wino<4, 2>(a, nws, st);
    if (W == 32) return mode == 0 ? launch_wino<5, 0>(a, nws, st) : launch_wino<5, 2>(a, nws, st);
    return mode == 0 ? launch_wino<6, 0>(a, nws, st) : launch_wino<6, 2>(a, nws, st);
}

// 1 if vf_wino_conv_fwd_gn can evaluate GroupNorm(groups) in the conv's fix-up launch at this shape (the sampler at a few
// views: every tile of the launch a K-split tail tile; 16x16 ... 64x64 maps)
int vf_wino_conv_gn_fusable(int S, int Cin, int Cout, int H, int W, int mode, int groups) {
    return (vf_wino_supported(H, W, mode) && wino_gn_fusable(S, Cin, Cout, H, W, groups)) ? 1 : 0;
}

// vf_wino_conv_fwd + a = [Swish](GroupNorm(groups, eps; gamma, beta)(y)) with the norm evaluated by the fix-up launch
// (reference unet.py:207-218: conv -> GroupNorm -> Swish of the next Block).  y is written only if store_y.
// hipErrorInvalidValue unless vf_wino_conv_gn_fusable().
int vf_wino_conv_fwd_gn(const float* x, const float* u_packed, const float* bias, const float* view_bias,
                        const float* residual, float* y, int store_y, const float* gn_gamma, const float* gn_beta,
                        float* a_out, int groups, float eps, int silu, float* ws, long ws_floats, int S, int Cin, int Cout,
                        int H, int W, int mode, void* stream) {
    if (S <= 0) return 0;
    if (!vf_wino_conv_gn_fusable(S, Cin, Cout, H, W, mode, groups) || !ws || !gn_gamma || !gn_beta || !a_out)
        return (int)hipErrorInvalidValue;
    WinoArgs a;
    a.x = x; a.u = u_packed; a.bias = bias; a.vbias = view_bias; a.res = residual; a.y = y;
    a.S = S; a.Cin = Cin; a.Cout = Cout; a.CinP = rup(Cin, WCK); a.CoutP = rup(Cout, WTCO);
    a.ws = ws;
    WinoGn gn;
    gn.gamma = gn_gamma; gn.beta = gn_beta; gn.a_out = a_out; gn.groups = groups; gn.silu = silu; gn.store_y = store_y; gn.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const size_t nws = (size_t)ws_floats;
    if (W == 16) return mode == 0 ? launch_wino<4, 0>(a, nws, st, gn) : launch_wino<4, 2>(a, nws, st, gn);
    if (W == 32) return mode == 0 ? launch_wino<5, 0>(a, nws, st, gn) : launch_wino<5, 2>(a, nws, st, gn);
    return mode == 0 ? launch_wino<6, 0>(a, nws, st, gn) : launch_wino<6, 2>(a, nws, st, gn);
}

}  // extern "C"

#ifdef VF_STAMPS
extern "C" void vf_debug_stamps(unsigned long long* out8, int reset) {
    if (out8) (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 128);
    if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, 128); }
}
#endif
