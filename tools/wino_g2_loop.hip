// Round-6 go / no-go microbenchmark for a second-generation F(4x4,3x3) forward kernel.
//
// Candidate structure ("two half workgroups"): workgroup = 4 waves = 64 co x 16 tiles (256 pixels), TWO independent
// workgroups per CU (one wave of each on every SIMD); wave w owns 16 co x 16 tiles x ALL 36 slices
// (v_mfma_f32_16x16x4_f32, 144 accumulators), so the output transform A4^T M A4 runs in registers -- no LDS exchange, no
// epilogue barriers -- and one workgroup's epilogue / prologue overlaps the other workgroup's chunk loop.
// What it costs: every workgroup streams the whole U block of its 64 channels for 16 tiles instead of 32 (2x the
// L2 -> CU bytes per multiply) and twice the MFMA instructions for the same flops.
//
// The benchmark runs the chunk loop with the real instruction mix on synthetic operands: per chunk (8 ci) and wave
// 72 MFMAs, 18 global_load_dwordx4 of U (a ring of nine slice pairs, half a chunk ahead), 18 ds_read_b128 of V,
// NVQ packed-fp32 instructions per slice pair (the window transform's share), 18 ds_write_b32, 3 row loads + 3 b128 LDS
// stores, one LDS-only barrier; per tile the in-register output transform + 16 float4 stores per lane.
// Reference points of the shipped kernel (tools/wino44f_stamps.py, S = 96): 5.5-5.8 k cycles per chunk per CU for 512
// pixels, 14.6 k cycles of epilogue + staging per 512-pixel tile: 60 k cycles per tile at Cin = 64, 150 k at Cin = 192.
// Gate: cycles per PAIR of 256-pixel tiles (two workgroups on a CU, one tile each) <= 48 k at 8 chunks.
//
// Build: hipcc --offload-arch=gfx950 -O3 -I view_fusion_amd/csrc tools/wino_g2_loop.hip -o tools/wino_g2_loop
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((address_space(1))) const f32x4 gf4;

constexpr int UCH = 36 * 64 * 8;                 // floats of one chunk's U block (64 co)
constexpr int VSZ = 18 * 64 * 4;                 // floats of one V buffer: [pair 18][lane 64][sl 2][j 2]

template <int NVQ, bool EPI, bool ULOAD>
__global__ __launch_bounds__(256, 2) void g2_loop(const float* __restrict__ U, const float* __restrict__ x,
                                                  float* __restrict__ y, unsigned long long* stamps, int nch,
                                                  int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[16128];          // 63 KB: two workgroups per CU
    float* const Vl = lds;                                             // [2][VSZ]
    float* const Pl = lds + 2 * VSZ;                                   // raw rows [2][8 ci][432]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16128; i += 256) lds[i] = (float)((i * 7) % 13) * 0.125f - 0.75f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();

    f32x2 dv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dv[i] = (f32x2){(float)(lane + i) * 1e-3f, (float)(lane - i) * 1e-3f};
    const f32x2 dk = (f32x2){0.999f, 1.001f};
    const char* ub = uniform_ptr(U);
    unsigned uoff = (unsigned)((wid * 64 + lane) * 16);                // bytes within a pair's 4 KB line
    const char* xb = uniform_ptr(x + (size_t)(blockIdx.x % 96) * 64 * 4096);
    unsigned xoff = (unsigned)(lane * 16 + wid * 4096 * 4);

    for (int tile = 0; tile < ntiles; ++tile) {
        f32x4 acc[36];
        f32x4 ur[9];
        f32x4 xr0, xr1, xr2;
#define ULD(SLOT, C, PP)                                                                                   \
    if (ULOAD) {                                                                                           \
        const char* b_ = ub + ((size_t)(C) * UCH + (PP) * 1024) * 4;                                       \
        asm("" : "+s"(b_), "+v"(uoff));                                                                    \
        ur[SLOT] = *(gf4*)((const __attribute__((address_space(1))) char*)b_ + uoff);                      \
    }
#pragma unroll
        for (int s = 0; s < 9; ++s) { ur[s] = (f32x4){0.5f, 0.25f, -0.5f, 0.125f}; ULD(s, 0, s); }
#define XLD(R, C, I)                                                                                       \
    {                                                                                                      \
        const char* b_ = xb + ((size_t)(C) * 8 * 4096 + (I) * 256) * 4;                                    \
        asm("" : "+s"(b_), "+v"(xoff));                                                                    \
        R = *(gf4*)((const __attribute__((address_space(1))) char*)b_ + xoff);                             \
    }
        XLD(xr0, 0, 0); XLD(xr1, 0, 1); XLD(xr2, 0, 2);
#pragma unroll
        for (int p = 0; p < 36; ++p) acc[p] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // one slice pair: B fragments of the pair, 4 MFMAs (two slices interleaved), ring reload, side work
#define PAIR(C, PAR, PP)                                                                                   \
    {                                                                                                      \
        const f32x4 b_ = *reinterpret_cast<const f32x4*>(Vl + (PAR) * VSZ + ((PP) * 64 + lane) * 4);       \
        const f32x4 u_ = ur[(PP) % 9];                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        acc[2 * (PP)] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_.x, b_.x, acc[2 * (PP)], 0, 0, 0);          \
        acc[2 * (PP) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_.z, b_.z, acc[2 * (PP) + 1], 0, 0, 0);  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int q = 0; q < NVQ / 2; ++q)                                                \
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(dv[(q + (PP)) & 7]) : "v"(dk));              \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        acc[2 * (PP)] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_.y, b_.y, acc[2 * (PP)], 0, 0, 0);          \
        acc[2 * (PP) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(u_.w, b_.w, acc[2 * (PP) + 1], 0, 0, 0);  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if ((PP) < 9) { ULD((PP) % 9, (C), (PP) + 9); } else { ULD((PP) % 9, (C) + 1, (PP) - 9); }         \
        _Pragma("unroll") for (int q = NVQ / 2; q < NVQ; ++q)                                              \
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(dv[(q + (PP)) & 7]) : "v"(dk));              \
        Vl[((PAR) ^ 1) * VSZ + (PP) * 256 + tid] = dv[(PP) & 7].x;                                         \
        if ((PP) >= 12 && (PP) < 15) {                                                                     \
            const f32x4 r_ = (PP) == 12 ? xr0 : (PP) == 13 ? xr1 : xr2;                                    \
            *reinterpret_cast<f32x4*>(Pl + (PAR) * 3456 + ((PP) - 12) * 1024 + tid * 4) = r_;              \
        }                                                                                                  \
        if ((PP) == 16) { XLD(xr0, (C) + 3, 0); XLD(xr1, (C) + 3, 1); XLD(xr2, (C) + 3, 2); }              \
    }
#define CHUNK(C, PAR)                                                                                      \
    {                                                                                                      \
        PAIR(C, PAR, 0) PAIR(C, PAR, 1) PAIR(C, PAR, 2) PAIR(C, PAR, 3) PAIR(C, PAR, 4) PAIR(C, PAR, 5)    \
        PAIR(C, PAR, 6) PAIR(C, PAR, 7) PAIR(C, PAR, 8) PAIR(C, PAR, 9) PAIR(C, PAR, 10) PAIR(C, PAR, 11)  \
        PAIR(C, PAR, 12) PAIR(C, PAR, 13) PAIR(C, PAR, 14) PAIR(C, PAR, 15) PAIR(C, PAR, 16)               \
        VF_LDS_BARRIER();                                                                                  \
        PAIR(C, PAR, 17)                                                                                   \
    }
        for (int c = 0; c < nch; c += 2) {
            CHUNK(c, 0);
            CHUNK(c + 1, 1);
        }
        if (EPI) {
            // output transform in registers: slices p = 6 a + b; co pairs (r, r + 1) on packed fp32
            float* const yo = y + ((size_t)(blockIdx.x * 4 + wid) * 16 + 4 * (lane >> 4)) * 4096 + 4 * (lane & 15);
#pragma unroll
            for (int rp = 0; rp < 2; ++rp) {
                f32x2 T[6][4];
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    f32x2 m[6];
#pragma unroll
                    for (int b = 0; b < 6; ++b) m[b] = (f32x2){acc[6 * a + b][2 * rp], acc[6 * a + b][2 * rp + 1]};
                    const f32x2 s12 = pk_add(m[1], m[2]), d12 = pk_sub(m[1], m[2]);
                    const f32x2 s34 = pk_add(m[3], m[4]), d34 = pk_sub(m[3], m[4]);
                    T[a][0] = pk_add(pk_add(m[0], s12), s34);
                    T[a][1] = pk_fmak<2>(d34, d12);
                    T[a][2] = pk_fmak<4>(s34, s12);
                    T[a][3] = pk_add(pk_fmak<8>(d34, d12), m[5]);
                }
                f32x2 Y[4][4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const f32x2 s12 = pk_add(T[1][c], T[2][c]), d12 = pk_sub(T[1][c], T[2][c]);
                    const f32x2 s34 = pk_add(T[3][c], T[4][c]), d34 = pk_sub(T[3][c], T[4][c]);
                    Y[0][c] = pk_add(pk_add(T[0][c], s12), s34);
                    Y[1][c] = pk_fmak<2>(d34, d12);
                    Y[2][c] = pk_fmak<4>(s34, s12);
                    Y[3][c] = pk_add(pk_fmak<8>(d34, d12), T[5][c]);
                }
#pragma unroll
                for (int yy = 0; yy < 4; ++yy) {
                    *reinterpret_cast<float4*>(yo + (size_t)(2 * rp) * 4096 + yy * 64) =
                        make_float4(Y[yy][0].x, Y[yy][1].x, Y[yy][2].x, Y[yy][3].x);
                    *reinterpret_cast<float4*>(yo + (size_t)(2 * rp + 1) * 4096 + yy * 64) =
                        make_float4(Y[yy][0].y, Y[yy][1].y, Y[yy][2].y, Y[yy][3].y);
                }
            }
            VF_LDS_BARRIER();                       // the next tile's staging barriers (two in the real kernel)
            VF_LDS_BARRIER();
        } else {
            float s = 0.f;
#pragma unroll
            for (int p = 0; p < 36; ++p) s += acc[p][0] + acc[p][1] + acc[p][2] + acc[p][3];
            y[(size_t)blockIdx.x * 256 + tid] = s;
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += dv[i].x + dv[i].y;
    if (s == 12345.f) y[0] = s;
    if (tid == 0) {
        atomicAdd(&stamps[0], __builtin_amdgcn_s_memtime() - t0);
        atomicAdd(&stamps[1], __builtin_amdgcn_s_memrealtime() - r0);
    }
}

template <int NVQ, bool EPI, bool ULOAD>
void run(const float* U, const float* x, float* y, unsigned long long* st, int nch, int ntiles, int blocks,
         const char* what) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    unsigned long long h[2] = {0, 0};
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipMemset(st, 0, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((g2_loop<NVQ, EPI, ULOAD>), dim3(blocks), dim3(256), 0, 0, U, x, y, st, nch, ntiles);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) { best = ms; (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost); }
    }
    const double cyc = (double)h[0] / blocks / ntiles;
    const double ghz = (double)h[0] / (double)h[1] * 0.1;
    const double tf = (double)blocks * ntiles * nch * 4 * 72 * 2048.0 / best / 1e9;
    printf("%-52s nch %2d  blocks %3d: %8.3f ms  %7.0f cycles per tile per workgroup (%5.0f per chunk)  %.2f GHz  %5.1f TF executed\n",
           what, nch, blocks, best, cyc, cyc / nch, ghz, tf);
}

int main(int argc, char** argv) {
    const int ntiles = 6;
    float *U, *x, *y; unsigned long long* st;
    const size_t nu = (size_t)UCH * 64, nx = (size_t)96 * 64 * 4096 + (1 << 20), ny = (size_t)512 * 64 * 4096;
    (void)hipMalloc(&U, nu * 4); (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&st, 16);
    std::vector<float> h(nu);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(U, h.data(), nu * 4, hipMemcpyHostToDevice);
    std::vector<float> hx(nx);
    for (auto& v : hx) v = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice);
    for (int nch : {8, 24}) {
        run<0, false, false>(U, x, y, st, nch, ntiles, 512, "MFMA + V reads + barrier only (no U, no VALU)");
        run<0, false, true>(U, x, y, st, nch, ntiles, 512, "+ U stream from L2");
        run<6, false, true>(U, x, y, st, nch, ntiles, 512, "+ 6 pk VALU per slice pair (108 / chunk)");
        run<6, true, true>(U, x, y, st, nch, ntiles, 512, "+ in-register output transform + stores");
        run<4, true, true>(U, x, y, st, nch, ntiles, 512, "same, 4 pk VALU per pair (72 / chunk)");
        run<6, true, true>(U, x, y, st, nch, ntiles, 256, "same as 6-VALU, ONE workgroup per CU (256 blocks)");
    }
    return 0;
}
