// Round-6 loop-level price of the path DESIGN 8-G names for the MFMA-bound part of the step: a DIRECT 3x3 convolution on
// bf16x3 split products (a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1), six v_mfma_f32_32x32x16_bf16 per 16 input
// channels, fp32 accumulation -- fp32-accurate by measurement, profiles/r03_bf16x3_experiment.md) for the 16x16 / 8x8 maps that
// run the nested Winograd kernel today (5.6 ms + 0.35 ms of fix-ups per training iteration at 0.60 matrix-pipe busy).
//
// The chunk loop with the instruction mix such a kernel would have, on synthetic operands; workgroup = 4 waves = 64 co x 256
// pixels (one 16x16 view), wave = 32 co x 128 pixels = four 32x32 accumulators; per 16-channel chunk and wave:
//   9 taps x { 3 A fragments (weights pre-split at pack time, 16 B per lane each, straight from L2, one tap ahead) ;
//              4 pixel tiles x { 3 B fragments by ds_read_b128 from [piece][k half][haloed pixel][8 ci bf16] ; 6 MFMAs } }
//   = 216 MFMAs (6 912 matrix cycles), 27 global_load_dwordx4, 108 ds_read_b128,
//   + staging of the next chunk: 16 dword loads, the three-way split of 16 values (v_cvt_pk_bf16_f32 + subtracts: ~72 vector
//     instructions), 6 ds_write_b128, one LDS-only barrier.
// Reported: cycles per chunk and workgroup against the 6 912 (one workgroup per CU) / 13 824 (two) of matrix issue, and
// what that would make of 192 -> 192 @ 16x16 at S = 96 (3 x 96 workgroup tiles x 12 chunks; nested Winograd today: 65.6 us).
// Build: hipcc --offload-arch=gfx950 -O3 -I view_fusion_amd/csrc tools/bf16x3_direct_loop.hip -o tools/bf16x3_direct_loop
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float bf_lo(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
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

constexpr int PIX = 18 * 18;                         // haloed 16x16 view
constexpr int PLANE = PIX * 4;                       // dwords of one [piece][k half] plane: 16 B per pixel
constexpr int BUF = 6 * PLANE;                       // one chunk: 3 pieces x 2 k halves = 31 104 B
constexpr int WCH = 9 * 3 * 2 * 64 * 4;              // dwords of one (co tile 64, chunk) weight block: [tap][piece][co half][lane][4]

// MODE bits: 1 = A fragments from global memory, 2 = B fragments from LDS, 4 = staging (loads + split + LDS stores) + barrier
template <int MODE>
__global__ __launch_bounds__(256, 2) void direct_loop(const unsigned* __restrict__ W, const float* __restrict__ x,
                                                      float* __restrict__ y, unsigned long long* stamps, int nch, int ntiles) {
    __shared__ __attribute__((aligned(16))) unsigned lds[2 * BUF];      // 62 208 B: two workgroups per CU
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cw = wid & 1, ph = wid >> 1;                               // co half, pixel half of this wave
    const int li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 0x3f803f80u + (unsigned)(i % 251);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    // lane -> pixel of a 32-pixel tile = two image rows of 16 (the b128 lane groups take 16 consecutive pixels each)
    const int prow = (li >> 4), pcol = li & 15;
    const unsigned* bbase = lds + lh * PLANE + ((1 + prow) * 18 + 1 + pcol) * 4;    // + pt * 2 rows + tap offset + piece
    const char* wb = uniform_ptr(W);
    unsigned woff = (unsigned)((cw * 64 + lane) * 16);
    const char* xb = uniform_ptr(x + (size_t)(blockIdx.x % 96) * 192 * 256);
    unsigned xoff = (unsigned)(((tid & 255)) * 4);                       // pixel
    float sx[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sx[i][j] = 0.25f * (float)(lane + j);

    for (int tile = 0; tile < ntiles; ++tile) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
        u32x4 af[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p) { af[0][p] = (u32x4){0x3f803f80u, 0x3f003f00u, 0x3e803e80u, 0x3f803f80u}; af[1][p] = af[0][p]; }
        for (int c = 0; c < nch; ++c) {
            const int buf = c & 1;
            if (MODE & 4) {                                             // the next chunk's activations: 16 channels x my pixel
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const char* b_ = xb + (size_t)((c * 16 + 8 * i + j) % 192) * 1024;
                        asm("" : "+s"(b_), "+v"(xoff));
                        sx[i][j] = *(const __attribute__((address_space(1))) float*)((const __attribute__((address_space(1))) char*)b_ + xoff);
                    }
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap & 1;
                if (MODE & 1) {                                         // A fragments of the NEXT tap, in the other register set
                    const char* b_ = wb + ((size_t)(c % 12) * WCH + ((tap + 1) % 9) * (3 * 2 * 64 * 4)) * 4;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const char* q_ = b_ + p * (2 * 64 * 16);
                        asm("" : "+s"(q_), "+v"(woff));
                        af[cur ^ 1][p] = *(const __attribute__((address_space(1))) u32x4*)((const __attribute__((address_space(1))) char*)q_ + woff);
                    }
                }
                const int toff = ((tap / 3 - 1) * 18 + (tap % 3 - 1)) * 4;
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    u32x4 bf[3];
                    if (MODE & 2) {
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            bf[p] = *reinterpret_cast<const u32x4*>(bbase + buf * BUF + p * 2 * PLANE + ((ph * 4 + pt) * 2 * 18) * 4 + toff);
                    } else {
#pragma unroll
                        for (int p = 0; p < 3; ++p) bf[p] = af[cur][p];
                    }
                    const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[cur][0]), a2 = __builtin_bit_cast(bf16x8, af[cur][1]),
                                 a3 = __builtin_bit_cast(bf16x8, af[cur][2]);
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, bf[0]), b2 = __builtin_bit_cast(bf16x8, bf[1]),
                                 b3 = __builtin_bit_cast(bf16x8, bf[2]);
                    f32x16 d = acc[pt];
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, d, 0, 0, 0);
                    acc[pt] = d;
                }
                if ((MODE & 4) && tap == 6) {                           // split + store the staged chunk into the other buffer
                    const int px = tid;                                 // my pixel of the view
                    unsigned* d_ = lds + (buf ^ 1) * BUF + ((1 + (px >> 4)) * 18 + 1 + (px & 15)) * 4;
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        u32x4 p1, p2, p3;
                        split8(sx[i], p1, p2, p3);
                        *reinterpret_cast<u32x4*>(d_ + (0 * 2 + i) * PLANE) = p1;
                        *reinterpret_cast<u32x4*>(d_ + (1 * 2 + i) * PLANE) = p2;
                        *reinterpret_cast<u32x4*>(d_ + (2 * 2 + i) * PLANE) = p3;
                    }
                }
            }
            if (MODE & 4) VF_LDS_BARRIER();
        }
        // epilogue: 64 accumulator registers -> 64 coalesced 128-byte row stores per wave
        float* yo = y + ((size_t)(blockIdx.x * 64 + cw * 32 + 4 * lh)) * 256 + ph * 128 + li;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) yo[(size_t)((r & 3) + 8 * (r >> 2)) * 256 + pt * 32] = acc[pt][r];
    }
    if (tid == 0) {
        atomicAdd(&stamps[0], __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&stamps[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int MODE>
void run(const unsigned* W, const float* x, float* y, unsigned long long* st, int nch, int ntiles, int blocks, const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(st, 0, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((direct_loop<MODE>), dim3(blocks), dim3(256), 0, 0, W, x, y, st, nch, ntiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost); }
    }
    const double cyc = (double)h[0] / blocks / ntiles / nch;
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double per_tile_us = best * 1e3 / ntiles;                         // one round of `blocks` workgroup tiles
    // 192 -> 192 @ 16x16, S = 96: 288 workgroup tiles of 12 chunks; `blocks` tiles run per round
    printf("%-46s blocks %3d: %7.3f ms  %6.0f cycles per chunk and workgroup  %.2f GHz  %5.1f us per round of tiles -> 288 tiles in %5.1f us (if they packed perfectly)\n",
           what, blocks, best, cyc, ghz, per_tile_us, per_tile_us * 288.0 / blocks);
}

int main() {
    const int nch = 12, ntiles = 6;
    unsigned* W; float *x, *y; unsigned long long* st;
    const size_t nw = (size_t)WCH * 12, nx = (size_t)96 * 192 * 256, ny = (size_t)512 * 64 * 256;
    (void)hipMalloc(&W, nw * 4); (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&st, 16);
    std::vector<unsigned> hw(nw);
    srand(1);
    for (auto& v : hw) v = 0x3f003f00u + (unsigned)(rand() & 0x007f007f);
    (void)hipMemcpy(W, hw.data(), nw * 4, hipMemcpyHostToDevice);
    std::vector<float> hx(nx);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    for (int blocks : {256, 512}) {
        run<0>(W, x, y, st, nch, ntiles, blocks, "MFMA only (216 per chunk and wave)");
        run<2>(W, x, y, st, nch, ntiles, blocks, "+ B fragments from LDS");
        run<3>(W, x, y, st, nch, ntiles, blocks, "+ A fragments from L2");
        run<7>(W, x, y, st, nch, ntiles, blocks, "+ staging, three-way split, barrier (= the loop)");
    }
    return 0;
}
