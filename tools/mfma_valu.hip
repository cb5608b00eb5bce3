// How much VALU work fits under fp32 MFMAs on gfx950?  Two waves per SIMD (one 512-thread workgroup per CU, as the
// Winograd kernels run); every wave loops over slices of four dependent v_mfma_f32_32x32x2_f32 (64 cycles each) with
// NV independent v_fma_f32 after each MFMA (sched_barrier keeps them there).  If VALU and MFMA co-issue freely the
// time stays flat until 2 waves x NV x 4 cycles reaches the 128 cycles two MFMAs take.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu.hip -o tools/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND, int AG>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = (f32x16){0};
    float a = src[lane], b = src[64 + lane];
    float v[32];
    typedef float f32x4q __attribute__((ext_vector_type(4)));
    f32x4q bq[2];                                        // 8 bf16 each (contents irrelevant here)
    bq[0] = (f32x4q){src[lane], src[lane + 1], src[lane + 2], src[lane + 3]};
    bq[1] = (f32x4q){src[lane + 4], src[lane + 5], src[lane + 6], src[lane + 7]};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 pk[9];
    int sc = 0;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = src[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 9; ++i) pk[i] = (f32x2){src[lane + i], src[lane + 2 * i]};
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = src[128 + i * 64 + lane];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 6; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                __builtin_amdgcn_sched_barrier(0);
                if (AG == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[s]) : "v"(bq[0]), "v"(bq[1]));  // bf16, 8 passes
                else if (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[s]) : "v"(a), "v"(b));   // accumulators in AGPRs
                else acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    if (KIND == 0) v[q & 31] = __builtin_fmaf(v[q & 31], 1.0001f, 0.5f);            // v_fma / v_fmac
                    else if (KIND == 1) v[q & 31] = v[q & 31] + v[(q + 1) & 31];                     // v_add, 2 sources
                    else if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc) :: "scc");               // SALU
                    else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[q & 7]) : "v"(pk[8]));  // packed fp32
                    else if (KIND == 4) v[q & 31] += lds[(lane + 64 * (q & 31) + it) & 4095];                  // ds_read_b32 + v_add
                    else if (KIND == 5) lds[(lane + 64 * (q & 31)) & 4095] = v[q & 31];                        // ds_write_b32
                    else if (KIND == 6) asm volatile("v_mov_b32 %0, %1" : "=v"(v[q & 31]) : "v"(a));           // v_mov
                }
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int i = 0; i < 9; ++i) s += pk[i].x + pk[i].y;
    s += (float)sc + lds[(lane * 7) & 4095];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV, int KIND, int AG = 0>
void run(const float* src, float* d) {
    const int blocks = 256, iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0, best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<NV, KIND, AG>), dim3(blocks), dim3(512), 0, 0, src, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double mf = (double)iters * 24;                 // MFMAs per wave
    // cycles per MFMA per SIMD at the achieved rate, assuming 2.4 GHz is NOT known -> report ns per (2 MFMAs) and TF
    const double tf = blocks * 8 * mf * 4096.0 / best / 1e9;
    printf("%s%s x %2d per MFMA: %8.3f ms  %6.1f TFLOP/s  (%.1f ns per MFMA pair)\n", AG == 2 ? "bf16 32x32x16 MFMA, " : AG ? "acc in AGPRs, " : "", KIND == 0 ? "v_fma " : KIND == 1 ? "v_add " : KIND == 2 ? "s_add " : KIND == 3 ? "v_pk_fma " : KIND == 4 ? "ds_read+v_add " : KIND == 5 ? "ds_write " : "v_mov ", NV, best, tf,
           best * 1e6 / mf);
}

int main() {
    float *src, *d;
    (void)hipMalloc(&src, 4096 * 4); (void)hipMalloc(&d, 256 * 512 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = (float)(i % 17) * 0.01f;
    (void)hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>(src, d); run<2, 0>(src, d); run<4, 0>(src, d); run<8, 0>(src, d); run<12, 0>(src, d); run<16, 0>(src, d); run<24, 0>(src, d); run<32, 0>(src, d);
    run<4, 1>(src, d); run<8, 1>(src, d); run<16, 1>(src, d); run<32, 1>(src, d);
    run<8, 2>(src, d); run<32, 2>(src, d); run<4, 3>(src, d); run<8, 3>(src, d); run<16, 3>(src, d);
    run<4, 4>(src, d); run<8, 4>(src, d); run<4, 5>(src, d); run<8, 5>(src, d); run<8, 6>(src, d); run<16, 6>(src, d);
    run<0, 1, 2>(src, d); run<4, 1, 2>(src, d); run<8, 1, 2>(src, d); run<16, 1, 2>(src, d);
    run<0, 1, 1>(src, d); run<4, 1, 1>(src, d); run<8, 1, 1>(src, d); run<16, 1, 1>(src, d); run<32, 1, 1>(src, d);
    return 0;
}
