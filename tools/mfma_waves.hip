// Does an MFMA + side-work loop behave the same with THREE waves per SIMD (768-thread workgroup, <= 170 VGPRs) as with
// two (512 threads)?  The question behind an F(4x4,3x3) forward kernel: 36 Winograd slices = 12 waves of
// 6 slices, per wave the same loop as today's 8-wave nested kernel (24 MFMAs + the same side work per 8-channel chunk).
// Every wave: iters x 24 dependent-by-accumulator v_mfma_f32_32x32x2_f32 over 6 accumulators, NV packed-fp32 VALU
// instructions + NL ds_read_b32 after each MFMA (sched_barrier keeps them there).  Reported: MFMA issue fraction
// (24 x 64 cycles x waves-per-SIMD / measured cycles per iteration and SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_waves.hip -o tools/mfma_waves
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NT, int NV, int NL>
__global__ __launch_bounds__(NT) void k(const float* __restrict__ src, float* out, long long* cyc, int iters) {
    const int lane = threadIdx.x & 63;
    __shared__ float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += NT) lds[i] = src[i & 4095];
    __syncthreads();
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = (f32x16){0};
    float a = src[lane], b = src[64 + lane], t = 0.f;
    f32x2 pk[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) pk[i] = (f32x2){src[lane + i], src[lane + 2 * i]};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __builtin_amdgcn_sched_barrier(0);
                acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NV; ++q) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[q & 7]) : "v"(pk[8]));
#pragma unroll
                for (int q = 0; q < NL; ++q) t += lds[(lane + 64 * (4 * s + e + q) + it) & 8191];
            }
        }
    }
    const long long t1 = clock64();
    float s = t;
#pragma unroll
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 9; ++i) s += pk[i].x + pk[i].y;
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;      // every wave: the slowest one counts
}

template <int NT, int NV, int NL>
void run(const float* src, float* d, long long* cyc) {
    const int blocks = 256, iters = 2000;
    static long long h[256 * 16];
    double best = 1e30;
    for (int rep = 0; rep < 4; ++rep) {
        hipLaunchKernelGGL((k<NT, NV, NL>), dim3(blocks), dim3(NT), 0, 0, src, d, cyc, iters);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double m = 0;
        for (int i = 0; i < 256; ++i) {
            long long mx = 0;
            for (int w = 0; w < NT / 64; ++w) mx = h[i * 16 + w] > mx ? h[i * 16 + w] : mx;
            m += (double)mx;
        }
        m /= 256;
        if (m < best) best = m;
    }
    const double per_iter = best / iters, wps = NT / 256.0;
    printf("%4d threads (%.0f waves/SIMD)  %2d v_pk_fma + %d ds_read per MFMA: %8.0f cycles per iteration, MFMA issue fraction %.3f\n",
           NT, wps, NV, NL, per_iter, 24 * 64 * wps / per_iter);
}

int main() {
    float *src, *d; long long* cyc;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&d, 256 * 1024 * 4); (void)hipMalloc(&cyc, 256 * 16 * 8);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 17) * 0.01f;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<512, 0, 0>(src, d, cyc); run<768, 0, 0>(src, d, cyc);
    run<512, 2, 1>(src, d, cyc); run<768, 2, 1>(src, d, cyc);
    run<512, 3, 2>(src, d, cyc); run<768, 3, 2>(src, d, cyc);
    run<512, 4, 2>(src, d, cyc); run<768, 4, 2>(src, d, cyc);
    run<512, 6, 3>(src, d, cyc); run<768, 6, 3>(src, d, cyc);
    return 0;
}
