// Sustained fp32-MFMA rate of this device: register-only v_mfma_f32_32x32x2_f32 loops on every CU.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
    float av = a + threadIdx.x * 1e-6f, bv = b + threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int wg_per_cu, float* d) {
    const int iters = 4000, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.5f, 0.25f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double fl = (double)blocks * 4 * iters * 8 * NACC * 4096.0;
        if (rep == 2) printf("NACC=%d wg/cu=%d : %.2f ms  %.1f TFLOP/s\n", NACC, wg_per_cu, ms, fl / ms / 1e9);
    }
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<1>(1, d); run<2>(1, d); run<4>(1, d); run<2>(2, d); run<2>(4, d); run<4>(2, d);
    // sustained: ~3 s of back-to-back launches, report the last
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL(k<4>, dim3(512), dim3(256), 0, 0, d, 20000, 0.5f, 0.25f);
    hipDeviceSynchronize();
    run<4>(2, d);
    return 0;
}
