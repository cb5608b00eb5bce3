// Micro-experiment: the Winograd kernel's MFMA phase in isolation: 16 accumulators (256 AGPRs),
// 4 dependent MFMAs per accumulator, 1 b128 + 4 b32 LDS reads per slice, one workgroup per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int ORDER>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float Ul[16 * 64 * 12];
    __shared__ __attribute__((aligned(16))) float Vl[16 * 8 * 64];
    for (int i = threadIdx.x; i < 16 * 64 * 12; i += 256) Ul[i] = (i % 97) * 1e-3f;
    for (int i = threadIdx.x; i < 16 * 8 * 64; i += 256) Vl[i] = (i % 89) * 1e-3f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const float* ub = Ul + ((wid & 1) * 32 + li) * 12 + 4 * lh;
    const float* vb = Vl + 4 * lh * 64 + (wid >> 1) * 32 + li;
    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = (f32x16){0};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (ORDER == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float4 a = *reinterpret_cast<const float4*>(ub + q * 64 * 12);
                float b[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) b[e] = vb[q * 8 * 64 + e * 64];
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[0], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[1], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[2], acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[3], acc[q], 0, 0, 0);
            }
        } else {   // two slices interleaved: consecutive MFMAs never share an accumulator
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const float4 a0 = *reinterpret_cast<const float4*>(ub + q * 64 * 12);
                const float4 a1 = *reinterpret_cast<const float4*>(ub + (q + 1) * 64 * 12);
                float b0[4], b1[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) { b0[e] = vb[q * 8 * 64 + e * 64]; b1[e] = vb[(q + 1) * 8 * 64 + e * 64]; }
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0[0], acc[q], 0, 0, 0);
                acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1[0], acc[q + 1], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0[1], acc[q], 0, 0, 0);
                acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1[1], acc[q + 1], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0[2], acc[q], 0, 0, 0);
                acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1[2], acc[q + 1], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0[3], acc[q], 0, 0, 0);
                acc[q + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1[3], acc[q + 1], 0, 0, 0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(t1 - t0) * 1e-30f;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (float)(t1 - t0) / (iters * 64.0f);
}
template <int ORDER>
void run(float* d) {
    const int iters = 200, blocks = 256;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<ORDER>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    float cyc; (void)hipMemcpy(&cyc, d + (1 << 20), 4, hipMemcpyDeviceToHost);
    printf("ORDER=%d: %.2f ms %.1f TFLOP/s, %.1f cycles per MFMA (wave 0)\n", ORDER, ms,
           (double)blocks * 4 * iters * 64 * 4096.0 / ms / 1e9, cyc);
}
int main() {
    float* d; (void)hipMalloc(&d, ((1 << 20) + 16) * 4);
    run<0>(d); run<1>(d);
    return 0;
}
