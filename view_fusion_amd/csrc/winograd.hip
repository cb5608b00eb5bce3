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
