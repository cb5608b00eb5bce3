// Fused Winograd F(2x2, 3x3) convolution for the stride-1 3x3 layers on the large feature maps
// (forward and dgrad; reference model/unet.py:42,189,214).  2.25x fewer multiplies than the direct
// form:   Y = A^T [ sum_ci (G g G^T) (.) (B^T d B) ] A   per 2x2 output tile.
// fp32 throughout; measured against the reference through the whole 33.9 M-parameter UNet the
// Winograd path deviates 3.4e-6 max-abs (direct: 1.4e-6), far inside the stated 5e-5 tolerance.
//
// Everything is fused in one kernel -- no transformed tensor ever goes to HBM:
//   * weights arrive pre-transformed + packed  U[co tile][chunk][k=16][co 64][ci 8]  (pack kernel);
//   * per 8-channel chunk the raw haloed input rows are staged in LDS, each thread transforms two
//     4x4 windows (B^T d B) into the V[k][ci][tile] LDS image;
//   * 16 independent GEMM slices  D_k[co][tile] += U_k[co][ci] V_k[ci][tile]  on
//     v_mfma_f32_32x32x2_f32 (A = U_k via one ds_read_b128 per 4 MFMAs, B = V_k with the tile on the
//     lane); a wave owns 32 co x 32 tiles x 16 k = 256 accumulators, ONE workgroup (4 waves =
//     64 co x 64 tiles = 256 output pixels) per CU with the whole register file;
//   * the epilogue applies A^T . A per (co, tile) in registers and stores 2x2 pixels per lane as
//     coalesced float2 rows, fusing bias + per-view bias + residual.
// MODE 0: plain input; MODE 2: nearest-x2-upsampled input (Upsample conv), as in conv.hip.
#include "common.h"

namespace {

constexpr int WTCO = 64;      // output channels per workgroup
constexpr int WTT = 64;       // 2x2 output tiles per workgroup
constexpr int WCK = 8;        // input channels per chunk

struct WinoArgs {
    const float* x;
    const float* u;       // packed transformed weights
    const float* bias;
    const float* vbias;
    const float* res;
    float* y;
    int S, Cin, Cout, CinP, CoutP;
    // tail splitting (small maps): workgroups [0, nfull) compute whole tiles; workgroup nfull + j*tail_split + p
    // computes the p-th K range of tile nfull + j and leaves a raw partial output in ws (wino_fixup_kernel)
    int nfull, tail_split;
    float* ws;
    int npers;            // persistent workgroups = min(nfull, WINO_PERSIST); blocks >= npers are the tail parts
};

constexpr int WINO_PERSIST = 256;     // one persistent workgroup per CU

template <int LOGW, int MODE>
struct WGeo {
    static constexpr int W = 1 << LOGW, H = W, HW = W * H, PAD = 1;
    static constexpr int SH = MODE == 2 ? H / 2 : H, SW = MODE == 2 ? W / 2 : W;   // source size
    static constexpr int TW = W / 2;                 // tiles per output row
    static constexpr int IPG = TW * TW >= WTT ? 1 : WTT / (TW * TW);   // images per workgroup (4 on 8x8 maps)
    static constexpr int TR = IPG == 1 ? WTT / TW : TW;                // tile rows per image in a workgroup
    static constexpr int WPI = IPG == 1 ? (H / 2) / TR : 1;            // workgroups per image (group)
    static constexpr int RPI = 2 * TR + 2;                             // haloed patch rows per image
    static constexpr int PH = IPG * RPI;
    // patch row: idx 3 = left halo, 4.. = pixels, 4+W = right halo.  With several images per workgroup the
    // row is W+4 wide and the right halo aliases the (never written, zero) idx 0 of the next row.
    static constexpr int PW = IPG == 1 ? W + 8 : W + 4;
    static constexpr int PS = PH * PW + (IPG == 1 ? 0 : 4);
    static constexpr int Q = W / 4;
    static_assert(TR >= 1 && WPI >= 1 && IPG * TR * TW == WTT, "unsupported map size for the 64-tile workgroup");
    // tile tl of a workgroup -> image in the group, tile row, tile column
    static __device__ __forceinline__ int t_img(int tl) { return tl / (TW * TR); }
    static __device__ __forceinline__ int t_row(int tl) { return (tl / TW) % TR; }
    static __device__ __forceinline__ int t_col(int tl) { return tl % TW; }
    // workgroup index (tile group) -> first view, first output row
    static __device__ __forceinline__ int g_view(int wg) { return IPG == 1 ? wg / WPI : wg * IPG; }
    static __device__ __forceinline__ int g_row(int wg) { return IPG == 1 ? (wg % WPI) * 2 * TR : 0; }
    static int groups(int S) { return IPG == 1 ? S * WPI : (S + IPG - 1) / IPG; }
};

// same patch addressing as conv.hip's load_patch4 (modes 0 and 2)
template <class G, int MODE>
__device__ __forceinline__ float4 wino_load4(const float* __restrict__ x, int S, int Cin, int s, int ci, int r0,
                                             int pr, int q) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s >= S || ci >= Cin) return v;
    const size_t plane = ((size_t)s * Cin + ci) * (size_t)(G::SH * G::SW);
    const int uy = r0 + pr - 1;
    if (uy < 0 || uy >= G::H) return v;
    if (MODE == 0) {
        v = *reinterpret_cast<const float4*>(x + plane + uy * G::SW + 4 * q);
    } else {
        const float2 t = *reinterpret_cast<const float2*>(x + plane + (uy >> 1) * G::SW + 2 * q);
        v = make_float4(t.x, t.x, t.y, t.y);
    }
    return v;
}

// Workgroup = 8 waves (two per SIMD): wave (cw, tw, kh) owns 32 channels x 32 tiles x the 8 slices of
// Winograd rows {2kh, 2kh+1} = 128 accumulators.  The two waves of a SIMD are the two row halves of
// the same block, so while one issues its staging / transform instructions the other keeps the
// matrix pipe busy.  The output transform is linear in the rows, so each half produces a partial
// 2x2 tile and the halves are summed through LDS once, in the epilogue.
//
// PERSISTENT over the whole tiles: the grid holds min(nfull, 256) workgroups (one per CU) and workgroup b computes the
// tiles b, b + 256, ... -- the order the hardware would have dispatched them in.  What that buys: the first global
// loads of the NEXT tile (U chunk 0, raw rows of chunks 0 and 1) are issued before the epilogue of the current one and
// land while it runs, instead of a fresh workgroup starting with a load round trip (~4.5k cycles of a ~52k-cycle tile
// on the 8-chunk 64->64 layers) with nothing else resident on its CU to cover it.  The K-split partial tiles of the
// tail plan stay one-shot workgroups behind the persistent ones.
template <int LOGW, int MODE>
__global__ __launch_bounds__(512, 2) void wino_conv_kernel(WinoArgs a) {
    using G = WGeo<LOGW, MODE>;
    constexpr int NT_ = 512;
    constexpr int NU4 = 16 * WTCO * 2;                   // float4 per U chunk
    constexpr int NUR = NU4 / NT_;                       // 4 per thread
    constexpr int NX4 = WCK * G::PH * G::Q;
    constexpr int NXR = (NX4 + NT_ - 1) / NT_;           // 1-2 per thread (tail predicated)
    constexpr int USZ = 16 * WTCO * WCK;                 // [k][co][ci 8], 16-B halves swizzled by the pack kernel
    constexpr int VSZ = 16 * WCK * WTT;
    constexpr int PSZ = WCK * G::PS;
    static_assert(NUR == 4, "U staging assumes 4 float4 per thread");

    // Everything is double buffered so that ONE barrier per chunk suffices: while the MFMAs of chunk c
    // read U[c&1] / V[c&1], the waves also write U(c+1), the raw rows of chunk c+2, and transform
    // the rows of chunk c+1 into V[(c+1)&1].
    __shared__ __attribute__((aligned(16))) float lds[2 * USZ + 2 * VSZ + 2 * PSZ];
    float* const Ul = lds;
    float* const Vl = lds + 2 * USZ;
    float* const Pl = lds + 2 * USZ + 2 * VSZ;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int cw = wid & 1, tw = (wid >> 1) & 1, kh = wid >> 2;
    const int li = lane & 31, lh = lane >> 5;
    const int ncot = a.CoutP / WTCO;
    const bool partial = (int)blockIdx.x >= a.npers;
    const int tail_id = partial ? (int)blockIdx.x - a.npers : 0;
    int c0 = 0, nch = a.CinP / WCK;                     // this workgroup's chunk range [c0, c0 + nch) (same for every
    if (partial) {                                      // tile of a persistent workgroup: the whole K)
        const int per = (nch + a.tail_split - 1) / a.tail_split;
        c0 = (tail_id % a.tail_split) * per;
        nch = max(0, min(nch - c0, per));
    }
    const int clast = max(nch - 1, 0);

    // per-thread staging layout of the raw input rows: tile independent (element e = tid + 512 i of the patch ->
    // (channel, patch row, float4 column)); only what the chunk loop touches is kept in registers
    int xlo[2], xci[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int e = tid + i * NT_;
        const int t1 = e / G::Q;
        xci[i] = t1 / G::PH;
        xlo[i] = xci[i] * G::PS + (t1 % G::PH) * G::PW + 4 * (e % G::Q) + 4;
    }

    // what changes from tile to tile and is needed inside the chunk loop: the weight / input bases and this thread's
    // source offsets (negative: row outside the image or the batch -> zeros).  The tile's position itself is
    // re-derived from its linear id where the epilogue needs it.
    struct Tile {
        const float* usrc;
        const float* xsrc;
        int xgo0, xgo1;
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
        t.usrc = a.u + ((size_t)cot * (a.CinP / WCK) + c0) * USZ;
        t.xsrc = a.x + (size_t)ts * a.Cin * (G::SH * G::SW);
        int go[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + i * NT_;
            const int q = e % G::Q, t1 = e / G::Q;
            const int pr = t1 % G::PH, ci = t1 / G::PH;
            const int img = pr / G::RPI;
            const int uy = tr0 + pr % G::RPI - 1;
            const bool ok = i < NXR && e < NX4 && ts + img < a.S && uy >= 0 && uy < G::H;
            go[i] = ok ? (img * a.Cin + ci) * (G::SH * G::SW) + (MODE == 0 ? uy * G::SW + 4 * q : (uy >> 1) * G::SW + 2 * q)
                       : -1;
        }
        t.xgo0 = go[0]; t.xgo1 = go[1];
        return t;
    };
    auto fetch_x = [&](const Tile& t, int i, int c) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int go = i == 0 ? t.xgo0 : t.xgo1;
        if (go >= 0 && (c0 + c) * WCK + xci[i] < a.Cin) {
            const float* p = t.xsrc + (size_t)(c0 + c) * WCK * (G::SH * G::SW) + go;
            if (MODE == 0) v = *reinterpret_cast<const float4*>(p);
            else { const float2 h = *reinterpret_cast<const float2*>(p); v = make_float4(h.x, h.x, h.y, h.y); }
        }
        return v;
    };

#ifdef VF_CONV_STAMPS   // diagnostic build only (tools/wino_stamps.py): clocks of the workgroup's FIRST tile
    long long st_[2] = {clock64(), 0}, rt0_ = wall_clock64();
#endif
    unsigned lin = partial ? 0u : blockIdx.x;           // linear id of the current whole tile
    const unsigned tail_logical = (unsigned)(a.nfull + tail_id / a.tail_split);
    Tile cur = make_tile(partial ? tail_logical : xcd_remap(lin, a.nfull));

    // named registers + macros (not arrays behind lambdas: those end up in scratch memory)
    float4 ur0, ur1, ur2, ur3;
    float4 xr0 = make_float4(0.f, 0.f, 0.f, 0.f), xr1 = xr0;
#define VF_ULOAD(T, I, C) ur##I = *reinterpret_cast<const float4*>((T).usrc + (size_t)(C) * USZ + 4 * (tid + (I) * NT_))
#define VF_USTORE(I, BUF) *reinterpret_cast<float4*>(Ul + (BUF) * USZ + 4 * (tid + (I) * NT_)) = ur##I
#define VF_ULOAD_ALL(T, C) { VF_ULOAD(T, 0, C); VF_ULOAD(T, 1, C); VF_ULOAD(T, 2, C); VF_ULOAD(T, 3, C); }
#define VF_USTORE_ALL(BUF) { VF_USTORE(0, BUF); VF_USTORE(1, BUF); VF_USTORE(2, BUF); VF_USTORE(3, BUF); }
#define VF_XLOAD(T, C) { xr0 = fetch_x((T), 0, (C)); if (NXR > 1) xr1 = fetch_x((T), 1, (C)); }
#define VF_XSTORE(BUF)                                                                                  \
    {                                                                                                   \
        if (tid < NX4) *reinterpret_cast<float4*>(Pl + (BUF) * PSZ + xlo[0]) = xr0;                      \
        if (NXR > 1 && tid + NT_ < NX4) *reinterpret_cast<float4*>(Pl + (BUF) * PSZ + xlo[1]) = xr1;     \
    }

    // input transform of this thread's 4x4 window (B^T d B): 512 windows = 8 channels x 64 tiles
    float d[16];
    const int wtl = tid & 63, wci = tid >> 6;
    const int wpo = wci * G::PS + (G::t_img(wtl) * G::RPI + 2 * G::t_row(wtl)) * G::PW + 2 * G::t_col(wtl) + 3;
    const int wvo = wci * WTT + wtl;
    auto win_read_row = [&](int r, int buf) {
        const float* p = Pl + buf * PSZ + wpo + r * G::PW;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r * 4 + c] = p[c];
    };
    auto win_write_row = [&](int r, int buf) {
        float* vo = Vl + buf * VSZ + wvo;
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = d[c], d1 = d[4 + c], d2 = d[8 + c], d3 = d[12 + c];
            t[c] = r == 0 ? d0 - d2 : (r == 1 ? d1 + d2 : (r == 2 ? d2 - d1 : d1 - d3));
        }
        vo[(4 * r + 0) * WCK * WTT] = t[0] - t[2];
        vo[(4 * r + 1) * WCK * WTT] = t[1] + t[2];
        vo[(4 * r + 2) * WCK * WTT] = t[2] - t[1];
        vo[(4 * r + 3) * WCK * WTT] = t[1] - t[3];
    };
    const int uoff = 8 * kh * WTCO * WCK + (cw * 32 + li) * WCK + 4 * (lh ^ ((li >> 4) & 1));
    const int voff = 8 * kh * WCK * WTT + 4 * lh * WTT + tw * 32 + li;

    // ---- first loads of the first tile: U(0), rows(0), rows(1) -- all three issued together (one round trip)
    VF_ULOAD_ALL(cur, 0);
    VF_XLOAD(cur, 0);
    float4 yr0 = fetch_x(cur, 0, min(1, clast)), yr1 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (NXR > 1) yr1 = fetch_x(cur, 1, min(1, clast));
    for (int i = tid; i < 2 * PSZ; i += NT_) Pl[i] = 0.f;    // halo columns stay zero in both buffers, for every tile

    for (;;) {
        f32x16 acc[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = (f32x16){0};

        // ---- prologue: U(0), rows(0), rows(1) staged; V(0) transformed; U(1), rows(2) in flight
        __syncthreads();                                  // zero fill done / previous tile's epilogue done with the LDS
        VF_USTORE_ALL(0);
        VF_XSTORE(0);
        if (tid < NX4) *reinterpret_cast<float4*>(Pl + PSZ + xlo[0]) = yr0;
        if (NXR > 1 && tid + NT_ < NX4) *reinterpret_cast<float4*>(Pl + PSZ + xlo[1]) = yr1;
        VF_ULOAD_ALL(cur, min(1, clast));
        VF_XLOAD(cur, min(2, clast));
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) win_read_row(r, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) win_write_row(r, 0);
        __syncthreads();

        // The staging code of chunks c+1..c+3 runs unconditionally with the chunk index clamped to the last one
        // (the final iterations redo harmless loads / LDS writes that nobody reads): without loop-tail branches
        // the compiler counts outstanding loads exactly; with them it falls back to s_waitcnt vmcnt(0) in front of
        // every staging access, i.e. a full load round trip per slice.
        for (int c = 0; c < nch; ++c) {
            const int cb = c & 1, nxt = cb ^ 1;
            const float* ub = Ul + cb * USZ + uoff;
            const float* vb = Vl + cb * VSZ + voff;
            float4 a_cur = *reinterpret_cast<const float4*>(ub);
            float b_cur[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) b_cur[e] = vb[e * WTT];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float4 a_nxt = a_cur;
                float b_nxt[4];
                // ---- side work of this slice, interleaved with the slice's own MFMAs: both waves of a SIMD run this
                // code in phase, so side work placed in front of the MFMAs leaves the matrix pipe idle in both at once
                // (measured: -1.35 ms per training step against "all side work, then four MFMAs").
                // Each staging register is stored to LDS in slice i and re-loaded (two chunks ahead) in slice
                // i+1: that leaves 7 of the 8 slices (~5000 cycles) between a global load and its use.
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur[0], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < 8) {                                              // operand fragments of the next slice
                    a_nxt = *reinterpret_cast<const float4*>(ub + (k + 1) * WTCO * WCK);
#pragma unroll
                    for (int e = 0; e < 4; ++e) b_nxt[e] = vb[(k + 1) * WCK * WTT + e * WTT];
                }
                if (k < 4) win_read_row(k, nxt);                              // rows of chunk c+1
                if (k >= 4) win_write_row(k - 4, nxt);
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur[1], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 0) VF_USTORE(0, nxt);
                if (k == 1) VF_USTORE(1, nxt);
                if (k == 2) VF_USTORE(2, nxt);
                if (k == 3) VF_USTORE(3, nxt);
                if (k == 4) VF_XSTORE(cb);                                    // rows of chunk c+2 -> buffer of chunk c
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur[2], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k == 1) VF_ULOAD(cur, 0, min(c + 2, clast));
                if (k == 2) VF_ULOAD(cur, 1, min(c + 2, clast));
                if (k == 3) VF_ULOAD(cur, 2, min(c + 2, clast));
                if (k == 4) VF_ULOAD(cur, 3, min(c + 2, clast));
                if (k == 5) VF_XLOAD(cur, min(c + 3, clast));
                __builtin_amdgcn_sched_barrier(0);
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur[3], acc[k], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (k + 1 < 8) {
                    a_cur = a_nxt;
#pragma unroll
                    for (int e = 0; e < 4; ++e) b_cur[e] = b_nxt[e];
                }
            }
            __syncthreads();
        }
#ifdef VF_CONV_STAMPS
        if (st_[1] == 0) st_[1] = clock64();
#endif

        const unsigned lin_next = lin + WINO_PERSIST;
        const bool has_next = !partial && lin_next < (unsigned)a.nfull;      // workgroup-uniform
        const unsigned logical_cur = partial ? tail_logical : xcd_remap(lin, a.nfull);
        // ---- output transform Y = A^T M A.  This wave holds rows {2kh, 2kh+1} of M (acc[4*(i-2kh)+j]):
        //   s0[j] = M0j+M1j (+M2j)      s1[j] = M1j (-M2j-M3j)     -> partial 2x2 tile, linear in the rows
        float part[16][4];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s0[4], s1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m0 = acc[j][r], m1 = acc[4 + j][r];          // rows 2kh, 2kh+1
                s0[j] = kh == 0 ? m0 + m1 : m0;
                s1[j] = kh == 0 ? m1 : -m0 - m1;
            }
            part[r][0] = s0[0] + s0[1] + s0[2];
            part[r][1] = s0[1] - s0[2] - s0[3];
            part[r][2] = s1[0] + s1[1] + s1[2];
            part[r][3] = s1[1] - s1[2] - s1[3];
        }
        // ---- the next whole tile of this (persistent) workgroup: its first loads go out now (the accumulators are
        // dead, so the staging registers are free) and land under the rest of the epilogue.  Issued unconditionally --
        // the last tile re-reads its own first chunks, which nobody uses -- so that the staging registers are dead
        // across the chunk loop instead of conditionally carried through it.
        const Tile nx = make_tile(has_next ? xcd_remap(lin_next, a.nfull) : logical_cur);
        VF_ULOAD_ALL(nx, 0);
        VF_XLOAD(nx, 0);
        yr0 = fetch_x(nx, 0, min(1, clast));
        if (NXR > 1) yr1 = fetch_x(nx, 1, min(1, clast));
        // The epilogue's per-lane index arithmetic is the same for every tile; left to itself the compiler hoists all of
        // it (~50 registers) out of the tile loop and then spills it across the chunk loop.  An opaque copy of the lane
        // id keeps it inside the epilogue, where the accumulators are dead and registers are plentiful.
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        const int li_e = lane_e & 31, lh_e = lane_e >> 5;
        float* xch = lds + (size_t)(wid & 3) * (64 * 64);                 // [value 64][lane 64] per (cw, tw) pair
        if (kh == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) xch[(r * 4 + q) * 64 + lane_e] = part[r][q];
        }
        __syncthreads();
        int s, r0, cot_;
        tile_pos(logical_cur, s, r0, cot_);
        const int co0 = cot_ * WTCO;
        if (kh == 0 && partial) {                            // raw partial tile: ws[tail_id][co 64][tile 64][2x2]
            const int tl = tw * 32 + li_e;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int col = cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e;
                float4 v;
                v.x = part[r][0] + xch[(r * 4 + 0) * 64 + lane_e];
                v.y = part[r][1] + xch[(r * 4 + 1) * 64 + lane_e];
                v.z = part[r][2] + xch[(r * 4 + 2) * 64 + lane_e];
                v.w = part[r][3] + xch[(r * 4 + 3) * 64 + lane_e];
                *reinterpret_cast<float4*>(a.ws + (((size_t)tail_id * WTCO + col) * WTT + tl) * 4) = v;
            }
        } else if (kh == 0 && s + G::t_img(tw * 32 + li_e) < a.S) {
            const int tl = tw * 32 + li_e;
            const int sv = s + G::t_img(tl);
            const int orow = r0 + 2 * G::t_row(tl), ocol = 2 * G::t_col(tl);
            // every epilogue operand is fetched BEFORE the first store: loads and stores retire through one
            // in-order counter, so a load issued after a store would wait for that store's round trip
            float eb[16], ev[16];
            float2 er[16][2];
            // (one uniform branch per operand kind, the 16 loads of a kind back to back: a branch between two
            // loads makes the compiler wait for the first before issuing the second)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                eb[r] = ev[r] = 0.f;
                er[r][0] = er[r][1] = make_float2(0.f, 0.f);
            }
            if (a.bias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e;
                    eb[r] = a.bias[min(co, a.Cout - 1)];
                }
            }
#ifndef VF_CONV_STAMPS
            if (a.vbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e;
                    ev[r] = a.vbias[(size_t)sv * a.Cout + min(co, a.Cout - 1)];
                }
            }
#endif
            if (a.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = min(co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e, a.Cout - 1);
                    const size_t o = ((size_t)sv * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
                    er[r][0] = *reinterpret_cast<const float2*>(a.res + o);
                    er[r][1] = *reinterpret_cast<const float2*>(a.res + o + G::W);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) eb[r] += ev[r];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + cw * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh_e;
                if (co >= a.Cout) continue;
                const size_t o = ((size_t)sv * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float2 v = make_float2(part[r][2 * i] + xch[(r * 4 + 2 * i) * 64 + lane_e] + eb[r] + er[r][i].x,
                                           part[r][2 * i + 1] + xch[(r * 4 + 2 * i + 1) * 64 + lane_e] + eb[r] + er[r][i].y);
                    *reinterpret_cast<float2*>(a.y + o + i * G::W) = v;
                }
            }
        }
#ifdef VF_CONV_STAMPS
        if (tid == 0 && a.bias == nullptr && a.vbias != nullptr && lin == blockIdx.x) {   // stamps ride in the vbias slot
            long long* o = reinterpret_cast<long long*>(const_cast<float*>(a.vbias)) + (size_t)blockIdx.x * 8;
            o[0] = st_[0]; o[1] = st_[1]; o[2] = clock64(); o[3] = 0; o[4] = 0; o[5] = st_[1] - st_[0];
            o[6] = rt0_; o[7] = wall_clock64();
        }
#endif
        if (!has_next) break;
        cur = nx;
        lin = lin_next;
    }
#undef VF_ULOAD
#undef VF_USTORE
#undef VF_ULOAD_ALL
#undef VF_USTORE_ALL
#undef VF_XLOAD
#undef VF_XSTORE
}

// OIHW -> transformed + packed forward  U[co tile][ci chunk][k][co 64][ci 8] = (G w G^T)_k
//        and backward (dgrad)          [ci tile][co chunk][k][ci 64][co 8] of the 180-degree-rotated kernel.
// One 512-thread workgroup per (tile, chunk) group = 8192 outputs: thread (m, k8) reads the nine taps of one
// (co, ci) pair once (36 contiguous bytes) and writes its 16 slices, each slice a contiguous 2 KB line of the
// workgroup.  Groups [0, nf/8192) are the forward pack, the rest the backward pack.
__device__ __forceinline__ void wino_pack_group(const float* __restrict__ w, float* __restrict__ uf,
                                                float* __restrict__ ub, int Cout, int Cin, size_t nf, size_t nb,
                                                size_t group) {
    const size_t ngf = nf / (16 * WTCO * WCK);
    const bool bwd = group >= ngf;
    if (bwd) {
        group -= ngf;
        if (group >= nb / (16 * WTCO * WCK) || !ub) return;
    }
    const int M = bwd ? Cin : Cout, K = bwd ? Cout : Cin;
    const int nchunk = (K + WCK - 1) / WCK;
    const int chunk = group % nchunk, mt = group / nchunk;
    const int t = threadIdx.x;
    const int m = t >> 3;
    const int k8 = (t & 7) ^ (((m >> 4) & 1) << 2);    // 16-B halves swapped on rows 16-31, 48-63: conflict-free ds_read_b128
    const int mm = mt * 64 + m, kk = chunk * WCK + k8;
    float g[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) g[i] = 0.f;
    if (mm < M && kk < K) {
        const int co = bwd ? kk : mm, ci = bwd ? mm : kk;
        const float* p = w + ((size_t)co * Cin + ci) * 9;
#pragma unroll
        for (int i = 0; i < 9; ++i) g[i] = bwd ? p[8 - i] : p[i];
    }
    // U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
    float tq[4][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const float h = 0.5f * (g[q] + g[6 + q]), e = 0.5f * g[3 + q];
        tq[0][q] = g[q];
        tq[1][q] = h + e;
        tq[2][q] = h - e;
        tq[3][q] = g[6 + q];
    }
    float* out = (bwd ? ub : uf) + group * (size_t)(16 * WTCO * WCK) + t;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float h = 0.5f * (tq[i][0] + tq[i][2]), e = 0.5f * tq[i][1];
        out[(4 * i + 0) * (WTCO * WCK)] = tq[i][0];
        out[(4 * i + 1) * (WTCO * WCK)] = h + e;
        out[(4 * i + 2) * (WTCO * WCK)] = h - e;
        out[(4 * i + 3) * (WTCO * WCK)] = tq[i][2];
    }
}

__global__ __launch_bounds__(512) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                        float* __restrict__ ub, int Cout, int Cin, size_t nf,
                                                        size_t nb) {
    wino_pack_group(w, uf, ub, Cout, Cin, nf, nb, blockIdx.x);
}

struct WPackDesc {
    const float* w;
    float* uf;
    float* ub;
    long long Cout, Cin, nf, nb, first_block;         // first_block in units of 256 outputs (32 per group)
};
__global__ __launch_bounds__(512) void wino_pack_multi_kernel(const WPackDesc* __restrict__ desc, int nlayers) {
    const long long vb = (long long)blockIdx.x * 32;
    int lo = 0, hi = nlayers;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (desc[mid].first_block <= vb) lo = mid; else hi = mid;
    }
    const WPackDesc d = desc[lo];
    wino_pack_group(d.w, d.uf, d.ub, (int)d.Cout, (int)d.Cin, (size_t)d.nf, (size_t)d.nb,
                    (size_t)((vb - d.first_block) / 32));
}

inline int rup(int v, int m) { return (v + m - 1) / m * m; }

// Sums the K-range partials of the tail tiles in a fixed order and applies the epilogue
// (bias + per-view bias + residual).  One thread per (tail tile, co, 2x2 tile).
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
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p = 0; p < a.tail_split; ++p) {
        const float4 t = *reinterpret_cast<const float4*>(
            a.ws + ((((size_t)j * a.tail_split + p) * WTCO + col) * WTT + tl) * 4);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    float b = 0.f;
    if (a.bias) b += a.bias[co];
    if (a.vbias) b += a.vbias[(size_t)s * a.Cout + co];
    const int orow = r0 + 2 * G::t_row(tl), ocol = 2 * G::t_col(tl);
    const size_t o = ((size_t)s * a.Cout + co) * G::HW + (size_t)orow * G::W + ocol;
    float2 v0 = make_float2(v.x + b, v.y + b), v1 = make_float2(v.z + b, v.w + b);
    if (a.res) {
        const float2 q0 = *reinterpret_cast<const float2*>(a.res + o);
        const float2 q1 = *reinterpret_cast<const float2*>(a.res + o + G::W);
        v0.x += q0.x; v0.y += q0.y; v1.x += q1.x; v1.y += q1.y;
    }
    *reinterpret_cast<float2*>(a.y + o) = v0;
    *reinterpret_cast<float2*>(a.y + o + G::W) = v1;
}

constexpr int WINO_SLOTS = 256;       // one 155 KB-LDS workgroup per CU

// How a grid of T equal tiles is finished when T is not a multiple of the slot count: the last R = T mod 256
// tiles are split over K into `split` parts each; the parts run in ceil(R*split/256) rounds of 1/split tile
// time.  `split` is the value in [1, min(8, nch/4)] that wastes the least CU time (ties: fewer parts).
inline double wino_tail_time(int R, int sp) { return (double)((R * sp + WINO_SLOTS - 1) / WINO_SLOTS) / sp; }

inline void wino_tail_plan(int T, int nch, int* nfull, int* split) {
    *nfull = T;
    *split = 1;
    const int R = T % WINO_SLOTS;
    if (R == 0 || T / WINO_SLOTS >= 3) return;          // tail round costs < 1/4 of the launch: leave it
    int best = 1;
    for (int sp = 2; sp <= 8 && sp <= nch / 4; ++sp) {    // at least 4 chunks per part
        const int per = (nch + sp - 1) / sp;              // the kernel gives each part `per` chunks:
        if ((nch + per - 1) / per != sp) continue;        // no part may start beyond the last chunk
        if (wino_tail_time(R, sp) < wino_tail_time(R, best) - 1e-9) best = sp;
    }
    if (best < 2) return;
    *nfull = T - R;
    *split = best;
}

template <int LOGW, int MODE>
int launch_wino(WinoArgs a, size_t ws_floats, hipStream_t st) {
    using G = WGeo<LOGW, MODE>;
    const int T = G::groups(a.S) * (a.CoutP / WTCO);
    wino_tail_plan(T, a.CinP / WCK, &a.nfull, &a.tail_split);
    const int ntail = T - a.nfull;
    if ((size_t)ntail * a.tail_split * WTCO * WTT * 4 > ws_floats || !a.ws) {   // no room: plain grid
        a.nfull = T;
        a.tail_split = 1;
    }
    const int nt = T - a.nfull;
    a.npers = a.nfull < WINO_PERSIST ? a.nfull : WINO_PERSIST;
    hipLaunchKernelGGL((wino_conv_kernel<LOGW, MODE>), dim3(a.npers + nt * a.tail_split), dim3(512), 0, st, a);
    if (nt > 0)
        hipLaunchKernelGGL((wino_fixup_kernel<LOGW>), dim3((nt * WTCO * WTT + 255) / 256), dim3(256), 0, st, a, nt);
    VF_RETURN_LAST_ERROR();
}

}  // namespace

extern "C" {

int vf_wino_pack_sizes(int Cout, int Cin, long* fwd_floats, long* bwd_floats) {
    *fwd_floats = 16L * rup(Cin, WCK) * rup(Cout, WTCO);
    *bwd_floats = 16L * rup(Cout, WCK) * rup(Cin, WTCO);
    return 0;
}

int vf_wino_pack_weights(const float* w_oihw, float* u_fwd, float* u_bwd, int Cout, int Cin, void* stream) {
    const size_t nf = 16UL * rup(Cin, WCK) * rup(Cout, WTCO);
    const size_t nb = u_bwd ? 16UL * rup(Cout, WCK) * rup(Cin, WTCO) : 0;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((nf + nb) / 8192)), dim3(512), 0, (hipStream_t)stream,
                       w_oihw, u_fwd, u_bwd, Cout, Cin, nf, nb);
    VF_RETURN_LAST_ERROR();
}

// desc: device int64 [nlayers][8] rows {w, u_fwd, u_bwd, Cout, Cin, fwd_floats, bwd_floats, first_block}
int vf_wino_pack_weights_multi(const void* desc, int nlayers, long total_blocks, void* stream) {
    if (nlayers <= 0 || total_blocks <= 0) return 0;
    hipLaunchKernelGGL(wino_pack_multi_kernel, dim3((unsigned)(total_blocks / 32)), dim3(512), 0, (hipStream_t)stream,
                       (const WPackDesc*)desc, nlayers);
    VF_RETURN_LAST_ERROR();
}

// 1 if vf_wino_conv_fwd supports this (output) size / mode: 3x3 stride 1, H = W in {8, 16, 32, 64}, modes 0 / 2.
int vf_wino_supported(int H, int W, int mode) {
    return (H == W && (W == 8 || W == 16 || W == 32 || W == 64) && (mode == 0 || mode == 2)) ? 1 : 0;
}

// workspace floats vf_wino_conv_fwd wants for its split tail tiles (0 when the grid divides evenly)
long vf_wino_conv_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const int tiles = (H / 2) * (W / 2);
    const int groups = tiles >= WTT ? S * (tiles / WTT) : (S + WTT / tiles - 1) / (WTT / tiles);
    const int T = groups * (rup(Cout, WTCO) / WTCO);
    int nfull, split;
    wino_tail_plan(T, rup(Cin, WCK) / WCK, &nfull, &split);
    return (long)(T - nfull) * split * WTCO * WTT * 4;
}

// Expected CU fill (percent) of vf_wino_conv_fwd at this shape under its tail plan, and the tile count;
// hosts use it to choose between this path and the direct kernel.
int vf_wino_conv_fill_pct(int S, int Cin, int Cout, int H, int W, int* tiles_out) {
    const int tiles = (H / 2) * (W / 2);
    const int groups = tiles >= WTT ? S * (tiles / WTT) : (S + WTT / tiles - 1) / (WTT / tiles);
    const int T = groups * (rup(Cout, WTCO) / WTCO);
    int nfull, split;
    wino_tail_plan(T, rup(Cin, WCK) / WCK, &nfull, &split);
    if (tiles_out) *tiles_out = T;
    if (T <= 0) return 0;
    const double time = (nfull + WINO_SLOTS - 1) / WINO_SLOTS + (T > nfull ? wino_tail_time(T - nfull, split) : 0.0);
    return (int)(100.0 * T / WINO_SLOTS / time);
}

// y = conv3x3(x) (+bias +view_bias +residual), pad 1, stride 1, via fused Winograd F(2x2,3x3).
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

}  // extern "C"

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
    // raw x staging duty: NX4 float4 (<= 2 per thread), e -> (ci, row, q) = (e / (NR*QPR), (e / QPR) % NR, e % QPR);
    // with HALO 64*4*2 = 512 halo scalars, one per thread: (ci, row, side) = (tid >> 3, (tid >> 1) & 3, tid & 1)
    float4 xr0, xr1;
    float xh;
    float2 dy0, dy1;
    auto chunk_pos = [&](int c, int& s, int& p, int& q0) {
        s = c / CPI;
        const int r = c - s * CPI;
        p = (r / CPR) * TR;                            // first tile row
        q0 = (r % CPR) * TPR;                          // first tile column
    };
    auto load_x = [&](int c) {
        int s, p, q0;
        chunk_pos(c, s, p, q0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 512 * i;
            const int ci = e / (NR * QPR), row = (e / QPR) % NR, q = e % QPR;
            const int gy = 2 * p - 1 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NX4 && ci0 + ci < a.Cin && gy >= 0 && gy < H) {
                if (MODE == 2) {
                    const float2 h = *reinterpret_cast<const float2*>(
                        a.x + (((size_t)s * a.Cin + ci0 + ci) * SH + (gy >> 1)) * SW + q0 + 2 * q);
                    v = make_float4(h.x, h.x, h.y, h.y);
                } else {
                    v = *reinterpret_cast<const float4*>(a.x + (((size_t)s * a.Cin + ci0 + ci) * H + gy) * W + 2 * q0 + 4 * q);
                }
            }
            if (i == 0) xr0 = v; else xr1 = v;
        }
        if (HALO) {
            const int ci = tid >> 3, row = (tid >> 1) & 3, side = tid & 1;
            const int gy = 2 * p - 1 + row, gx = side ? 2 * q0 + 2 * TPR : 2 * q0 - 1;
            xh = 0.f;
            if (ci0 + ci < a.Cin && gy >= 0 && gy < H && gx >= 0 && gx < W)
                xh = MODE == 2 ? a.x[(((size_t)s * a.Cin + ci0 + ci) * SH + (gy >> 1)) * SW + (gx >> 1)]
                               : a.x[(((size_t)s * a.Cin + ci0 + ci) * H + gy) * W + gx];
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = tid + 512 * i;
            if (e < NX4)
                *reinterpret_cast<float4*>(Xl + (e / (NR * QPR)) * XCS + ((e / QPR) % NR) * GXW + 4 + 4 * (e % QPR)) = i == 0 ? xr0 : xr1;
        }
        if (HALO) Xl[(tid >> 3) * XCS + ((tid >> 1) & 3) * GXW + ((tid & 1) ? 4 + 2 * TPR : 3)] = xh;
    };
    if (!HALO) {                                       // the strip spans the map: both halo columns are padding
        for (int e = tid; e < 64 * NR * 2; e += 512)
            Xl[(e / (2 * NR)) * XCS + ((e >> 1) % NR) * GXW + ((e & 1) ? 4 + 2 * TPR : 3)] = 0.f;
    }
    auto load_dy = [&](int c) {
        int s, p, q0;
        chunk_pos(c, s, p, q0);
        dy0 = dy1 = make_float2(0.f, 0.f);
        if (co0 + tch < a.Cout) {
            const float* g = a.dy + (((size_t)s * a.Cout + co0 + tch) * H + 2 * (p + ttr)) * W + 2 * (q0 + ttc);
            dy0 = *reinterpret_cast<const float2*>(g);
            dy1 = *reinterpret_cast<const float2*>(g + W);
        }
    };
    float bias_acc = 0.f;                              // sum of this thread's dY tiles (the bias gradient rides along)
    auto xform_dy = [&](int buf) {                     // dM = A dY A^T, A^T = [[1,1,1,0],[0,1,-1,-1]]
        bias_acc += (dy0.x + dy0.y) + (dy1.x + dy1.y);
        const float r[4][2] = {{dy0.x, dy0.y}, {dy0.x + dy1.x, dy0.y + dy1.y}, {dy0.x - dy1.x, dy0.y - dy1.y},
                               {-dy1.x, -dy1.y}};
        float* mo = Ml + buf * MSZ + tch * GT + tsw;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            mo[(4 * i + 0) * 64 * GT] = r[i][0];
            mo[(4 * i + 1) * 64 * GT] = r[i][0] + r[i][1];
            mo[(4 * i + 2) * 64 * GT] = r[i][0] - r[i][1];
            mo[(4 * i + 3) * 64 * GT] = -r[i][1];
        }
    };
    float d[16];
    auto xform_x_read = [&](int r) {
        const float* p = Xl + tch * XCS + (2 * ttr + r) * GXW + 2 * ttc + 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r * 4 + c] = p[c];
    };
    auto xform_x_write = [&](int r, int buf) {         // V = B^T d B
        float* vo = Vl + buf * MSZ + tch * GT + tsw;
        float t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float d0 = d[c], d1 = d[4 + c], d2 = d[8 + c], d3 = d[12 + c];
            t[c] = r == 0 ? d0 - d2 : (r == 1 ? d1 + d2 : (r == 2 ? d2 - d1 : d1 - d3));
        }
        vo[(4 * r + 0) * 64 * GT] = t[0] - t[2];
        vo[(4 * r + 1) * 64 * GT] = t[1] + t[2];
        vo[(4 * r + 2) * 64 * GT] = t[2] - t[1];
        vo[(4 * r + 3) * 64 * GT] = t[1] - t[3];
    };

    f32x16 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (f32x16){0};

    if (n > 0) {
        // ---- prologue: chunk 0 transformed into buffer 0; chunk 1 raw data in registers
        load_x(c_begin);
        load_dy(c_begin);
        store_x();
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) xform_x_read(r);
#pragma unroll
        for (int r = 0; r < 4; ++r) xform_x_write(r, 0);
        xform_dy(0);
        if (n > 1) { load_x(c_begin + 1); load_dy(c_begin + 1); }
        __syncthreads();
    }

    const int aoff = 8 * kh * 64 * GT + (cw * 32 + li) * GT + 4 * (lh ^ ((li >> 4) & 1));
    const int boff = 8 * kh * 64 * GT + (ciw * 32 + li) * GT + 4 * (lh ^ ((li >> 4) & 1));
    for (int c = 0; c < n; ++c) {
        const int cur = c & 1, nxt = cur ^ 1;
        const bool has1 = c + 1 < n, has2 = c + 2 < n;
        const float* ab = Ml + cur * MSZ + aoff;
        const float* bb = Vl + cur * MSZ + boff;
        float4 a_cur = *reinterpret_cast<const float4*>(ab);
        float4 b_cur = *reinterpret_cast<const float4*>(bb);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float4 a_nxt = a_cur, b_nxt = b_cur;
            // ---- side work (raw strip of chunk c+1 -> LDS in slice 0, barrier, transforms in slices 4-7; spreading
            // them over slices 2-6 measured the same),
            // interleaved with the slice's own MFMAs: the two waves of a SIMD run in phase, side work in front of
            // the MFMAs would idle the matrix pipe in both at once
            __builtin_amdgcn_sched_barrier(0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < 8) {
                a_nxt = *reinterpret_cast<const float4*>(ab + (k + 1) * 64 * GT);
                b_nxt = *reinterpret_cast<const float4*>(bb + (k + 1) * 64 * GT);
            }
            if (k == 0 && has1) store_x();
            if (k == 1 && has2) load_x(c_begin + c + 2);
            if (k == 4) __syncthreads();               // raw strip of chunk c+1 visible to every thread
            if (has1) {
                if (k == 4) xform_x_read(0);
                if (k == 5) xform_x_read(2);
                if (k == 6) xform_x_write(0, nxt);
                if (k == 7) xform_x_write(2, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (has1) {
                if (k == 4) xform_x_read(1);
                if (k == 5) xform_x_read(3);
                if (k == 6) xform_x_write(1, nxt);
                if (k == 7) xform_x_write(3, nxt);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k == 5 && has1) xform_dy(nxt);
            if (k == 6 && has2) load_dy(c_begin + c + 2);
            __builtin_amdgcn_sched_barrier(0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            a_cur = a_nxt; b_cur = b_nxt;
        }
        __syncthreads();
    }

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
    for (int k = 0; k < 16; ++k) u[k] = p[k * kstride];
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
long vf_wino_wgrad_ws_floats(int S, int Cin, int Cout, int H, int W) {
    const long slab = 16L * rup(Cout, 64) * rup(Cin, 32);
    const int nco = rup(Cout, 64) / 64, nci = (rup(Cin, 32) + 63) / 64;
    long z = 256 / (nco * nci);
    if (z < 1) z = 1;
    const long nchunks = (long)S * (H / 2) * (W / 2) / GT;
    if (z > nchunks) z = nchunks;
    return (z + 1) * slab + 256L * rup(Cout, 64);
}

int vf_wino_wgrad_supported(int H, int W, int mode) {
    return H == W && (W == 8 || W == 16 || W == 32 || W == 64) && (mode == 0 || mode == 2);
}

// dw[Cout][Cin][3][3] of a stride-1 3x3 conv (H = W = output size in {8, 16, 32, 64}; mode 2: x is stored at half
// size and nearest-upsampled on read) via Winograd F(2x2,3x3)
// db (or NULL): also the bias gradient sum_{s,p} dY[s][co][p] -- the kernel reads every dY tile anyway;
// db2 (or NULL): a second [Cout] destination for the same sums (the residual 1x1 conv shares this dY)
int vf_wino_wgrad(const float* x, const float* dy, float* dw, float* db, float* db2, float* ws, long ws_floats, int S,
                  int Cin, int Cout, int H, int W, int mode, void* stream) {
    if (S <= 0) return 0;
    if (!vf_wino_wgrad_supported(H, W, mode)) return (int)hipErrorInvalidValue;
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
