// Micro-experiment: fp32 MFMA fed from LDS in the conv kernel's pattern (1 b128 + 8 b32 per 8 MFMAs),
// no global traffic, no barriers in the loop.  What fraction of the register-only rate survives?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float wl[9 * 64 * 12];
    __shared__ __attribute__((aligned(16))) float xl[8 * 4 * 72];
    for (int i = threadIdx.x; i < 9 * 64 * 12; i += 256) wl[i] = i * 1e-5f;
    for (int i = threadIdx.x; i < 8 * 4 * 72; i += 256) xl[i] = i * 1e-5f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const float* wb = wl + ((wid & 1) * 32 + li) * 12 + 4 * lh;
    const int xo0 = 4 * lh * 288 + (wid >> 1) * 72 + li + 3, xo1 = xo0 + 32;
    f32x16 a0 = {0}, a1 = {0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 9; ++g) {
            const float4 av = *reinterpret_cast<const float4*>(wb + g * 64 * 12);
            const int off = (g / 3) * 72 + (g % 3);
            float b0[4], b1[4];
            if (VAR == 0) {
#pragma unroll
                for (int s = 0; s < 4; ++s) { b0[s] = xl[xo0 + off + s * 288]; b1[s] = xl[xo1 + off + s * 288]; }
            } else {   // VAR 1: B operands from registers (only the A b128 read remains)
#pragma unroll
                for (int s = 0; s < 4; ++s) { b0[s] = av.x + s; b1[s] = av.y + s; }
            }
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1[0], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0[1], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1[2], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0[3], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1[3], a1, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int VAR>
void run(int wg_per_cu, float* d) {
    const int iters = 400, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    }
    double fl = (double)blocks * 4 * iters * 72 * 4096.0;
    printf("VAR=%d wg/cu=%d : %.2f ms  %.1f TFLOP/s\n", VAR, wg_per_cu, ms, fl / ms / 1e9);
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; ++w) run<0>(w, d);
    for (int w = 1; w <= 4; ++w) run<1>(w, d);
    return 0;
}
